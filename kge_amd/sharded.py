"""Entity-sharded scoring and ranking over several GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; the CPU tests use
"gloo").  The reference has no distributed code at all; what makes the path shardable is in
the reference itself: score columns are independent and EntityRankingJob sums per-chunk
(rank, ties) counts over disjoint entity ranges (kge/job/eval_entity_ranking.py:222-313).
"One entity chunk per GPU + all-reduce(sum) of the int64 counters" therefore reproduces the
unsharded ranks exactly.

Layout: rank g owns entity rows [g*S, min((g+1)*S, E)), S = ceil(E / G); the relation table
(<= ~1 MB) is replicated.  Exchange steps, all small (latency-bound, <= ~1 MB):
  1. query rows: every rank fills the rows it owns, zeros elsewhere, ONE all-reduce(sum)
     (x + 0 is exact, so the gathered rows are bit-identical to the owner's);
  2. true scores: computed by the owner of the true entity, zeros elsewhere, all-reduce(sum);
  3. rank/tie counters: int64 all-reduce(sum).
Scoring itself needs no collective: each rank writes its own [n, E_g] slab.
"""
from typing import Optional

import torch
import torch.distributed as dist


class ShardedEntityTable:
    def __init__(self, scorer: str, ent_local: torch.Tensor, rel: torch.Tensor, num_entities: int,
                 l_norm: float = 1.0, group=None, backend=None):
        self.scorer, self.l_norm = scorer, float(l_norm)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.E = int(num_entities)
        self.shard = (self.E + self.world - 1) // self.world
        self.lo = min(self.rank * self.shard, self.E)
        self.hi = min(self.lo + self.shard, self.E)
        if ent_local.shape[0] != self.hi - self.lo:
            raise ValueError(f"rank {self.rank} must hold rows [{self.lo},{self.hi}) of the entity table")
        self.ent_local, self.rel = ent_local, rel
        if backend is None:
            from . import engine as backend  # the HIP kernels; no CPU fallback
        self.backend = backend

    @staticmethod
    def partition(num_entities: int, world: int, rank: int):
        shard = (num_entities + world - 1) // world
        lo = min(rank * shard, num_entities)
        return lo, min(lo + shard, num_entities)

    # ---- exchange steps -------------------------------------------------------------------
    def _allreduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def gather_entity_rows(self, idx: torch.Tensor) -> torch.Tensor:
        """[n, d] rows of the GLOBAL entity table for global ids `idx` (exchange step 1)."""
        idx = idx.long()
        own = (idx >= self.lo) & (idx < self.hi)
        rows = torch.zeros(idx.numel(), self.ent_local.shape[1], dtype=self.ent_local.dtype,
                           device=self.ent_local.device)
        rows[own] = self.ent_local[idx[own] - self.lo]
        return self._allreduce(rows)

    # ---- scoring: local slabs, no collective ---------------------------------------------
    def score_sp(self, s: torch.Tensor, p: torch.Tensor, s_rows: Optional[torch.Tensor] = None):
        """[n, E_g]: scores of (s_i, p_i, ·) against this rank's entities."""
        s_rows = self.gather_entity_rows(s) if s_rows is None else s_rows
        return self.backend.score_emb(self.scorer, s_rows, self.rel[p.long()], self.ent_local, "sp_",
                                      self.l_norm)

    def score_po(self, p: torch.Tensor, o: torch.Tensor, o_rows: Optional[torch.Tensor] = None):
        o_rows = self.gather_entity_rows(o) if o_rows is None else o_rows
        return self.backend.score_emb(self.scorer, self.ent_local, self.rel[p.long()], o_rows, "_po",
                                      self.l_norm)

    def score_sp_po(self, s: torch.Tensor, p: torch.Tensor, o: torch.Tensor):
        """[n, 2 E_g]: score_sp and score_po slabs side by side from one launch (backends with
        score_emb_sp_po), else the two calls."""
        s_rows, o_rows = self.gather_entity_rows(s), self.gather_entity_rows(o)
        if hasattr(self.backend, "score_emb_sp_po"):
            return self.backend.score_emb_sp_po(self.scorer, s_rows, self.rel[p.long()], o_rows, self.ent_local,
                                                self.l_norm)
        return torch.cat([self.score_sp(s, p, s_rows), self.score_po(p, o, o_rows)], dim=1)

    def true_scores(self, slab: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """Score of each row's true entity, taken from the owner's slab (exchange step 2)."""
        target = target.long()
        own = (target >= self.lo) & (target < self.hi)
        t = torch.zeros(target.numel(), dtype=torch.float32, device=slab.device)
        r = torch.nonzero(own).view(-1)
        t[r] = slab[r, target[r] - self.lo]
        return self._allreduce(t)

    def rank_counts(self, slab, true, lbl_rowptr=None, lbl_col=None, true_col=None, atol=1e-5,
                    rtol=1e-4):
        """Global (rank, ties) of every row: local counts over this shard's columns (labels and
        true_col are GLOBAL entity ids; the shard offset is applied by the kernel), then the
        int64 all-reduce (exchange step 3)."""
        rank, ties = self.backend.rank_counts(slab, true, lbl_rowptr, lbl_col, self.lo, true_col,
                                              atol, rtol)
        both = torch.stack([rank, ties])
        self._allreduce(both)
        return both[0], both[1]

    def rank_batch(self, triples: torch.Tensor, labels=None, atol=1e-5, rtol=1e-4):
        """(s_rank, s_ties, o_rank, o_ties) for a batch of (s,p,o) triples; `labels` =
        (sp_rowptr, sp_col, po_rowptr, po_col) CSR of filtered GLOBAL entity ids or None."""
        s, p, o = triples[:, 0], triples[:, 1], triples[:, 2]
        sp = self.score_sp(s, p)
        po = self.score_po(p, o)
        o_true = self.true_scores(sp, o)
        s_true = self.true_scores(po, s)
        sp_rp, sp_col, po_rp, po_col = labels if labels is not None else (None,) * 4
        o_rank, o_ties = self.rank_counts(sp, o_true, sp_rp, sp_col, o.long(), atol, rtol)
        s_rank, s_ties = self.rank_counts(po, s_true, po_rp, po_col, s.long(), atol, rtol)
        return s_rank, s_ties, o_rank, o_ties

    def rank_batch_multi(self, triples: torch.Tensor, filters_o, filters_s, atol=1e-5, rtol=1e-4):
        """Raw + len(filters) filtered rankings of a batch from ONE scan per direction and ONE
        counter all-reduce: filters_o / filters_s = [(begin [n], end [n], values), ...] ranges
        into a filter index's value arrays (GLOBAL entity ids; kge_filter_lookup /
        FilterIndex.ranges) for the sp_ / _po direction.  Returns int64 counts
        [2 (o, s), 2 (rank, ties), len(filters) + 1, n]."""
        s, p, o = triples[:, 0], triples[:, 1], triples[:, 2]
        n, K = triples.shape[0], len(filters_o)
        both = self.score_sp_po(s, p, o)
        c = both.shape[1] // 2
        sp, po = both[:, :c], both[:, c:]
        o_true = self.true_scores(sp, o)
        s_true = self.true_scores(po, s)
        counts = torch.zeros(2, 2, K + 1, n, dtype=torch.int64, device=sp.device)
        self.backend.rank_counts_multi(sp, o_true, filters_o, self.lo, o.long().contiguous(), atol, rtol,
                                       counts[0, 0], counts[0, 1])
        self.backend.rank_counts_multi(po, s_true, filters_s, self.lo, s.long().contiguous(), atol, rtol,
                                       counts[1, 0], counts[1, 1])
        return self._allreduce(counts)

    def topk(self, slab: torch.Tensor, k: int):
        """Global top-k (scores, entity ids) per row: local top-k, all-gather, merge
        (north_star's "RCCL all-gather of per-shard top-k")."""
        kk = min(k, slab.shape[1])
        v, i = torch.topk(slab, kk, dim=1)
        i = i + self.lo
        if kk < k:  # pad short shards
            pad = k - kk
            v = torch.cat([v, torch.full((v.shape[0], pad), float("-inf"), device=v.device)], 1)
            i = torch.cat([i, torch.full((i.shape[0], pad), -1, dtype=i.dtype, device=i.device)], 1)
        if self.world > 1:
            vs = [torch.empty_like(v) for _ in range(self.world)]
            is_ = [torch.empty_like(i) for _ in range(self.world)]
            dist.all_gather(vs, v, group=self.group)
            dist.all_gather(is_, i, group=self.group)
            v, i = torch.cat(vs, 1), torch.cat(is_, 1)
        tv, ti = torch.topk(v, k, dim=1)
        return tv, torch.gather(i, 1, ti)
