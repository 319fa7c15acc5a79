"""Entity-sharded scoring and ranking over several GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; the CPU tests use
"gloo").  The reference has no distributed code at all; what makes the path shardable is in
the reference itself: score columns are independent and EntityRankingJob sums per-chunk
(rank, ties) counts over disjoint entity ranges (kge/job/eval_entity_ranking.py:222-313).
"One entity chunk per GPU + all-reduce(sum) of the int64 counters" therefore reproduces the
unsharded ranks exactly.

Layout: rank g owns entity rows [g*S, min((g+1)*S, E)), S = ceil(E / G); the relation table
(<= ~1 MB) is replicated; every rank sees the same batch of queries.  Exchange steps, all small
(latency-bound, <= ~2 MB) and all free of host synchronisation (no boolean-mask indexing, no
nonzero: every shape is fixed by n, so a step can be captured in a hipGraph):
  1. query rows: ONE gather launch (kge_embed) fills a fixed [k*n, d] send block
     -- row i from this shard if it owns id i, any local row otherwise -- ONE
     all_gather_into_tensor, and one more gather launch picks row i out of its owner's block
     (index owner(i)*k*n + i, computed on the device).  The s and the o rows of a batch travel in
     the same collective; the relation rows come from the replicated table in the first launch;
  2. true scores: taken from the owner's slab (torch.where on the ownership mask), zeros
     elsewhere, all-reduce(sum) of n floats (x + 0 is exact);
  3. rank/tie counters: one int64 all-reduce(sum) for all rankings of both directions.
Scoring itself needs no collective: each rank writes its own [n, E_g] slab.

Training (1vsAll with the kl loss = cross entropy over ALL entities, train_1vsAll.py:64-81), SURVEY 8e
(3)-(4): `ce_loss` runs the fused score + loss kernels per shard on the exchanged query rows
(kge_ce_emb_fwd: the shard's log-sum-exp and, on the owner, the label's score), merges the shards'
log-sum-exps (all-gather of n floats, logsumexp) and the owner's score (all-reduce of n floats), and in
the backward hands the GLOBAL log-sum-exp to kge_ce_emb_bwd: softmax - onehot over the shard's columns,
both gradient products per shard; the query-row gradients are summed over the shards (ONE all-reduce of
[n, d + d_r] floats) and scatter-added by the owners; the gradient of a shard's own rows never leaves
it and the relation gradients come out identical on every rank (no E x d all-reduce, no relation
all-reduce).  Optimizer state follows the rows: every rank steps its own shard.
"""
from typing import Optional

import os

import torch
import torch.distributed as dist


class _ShardedCE(torch.autograd.Function):
    """Per-row cross entropy of the batch's sp_ (or _po) scores over the entities of ALL shards."""

    @staticmethod
    def forward(ctx, sh, direction, ent_master, rel_master, ids, p, labels):
        rows, rel_rows = sh.exchange_rows([ids], p)
        rows, rel_rows = rows.clone(), rel_rows.clone()  # the exchange buffers are reused by the next call
        lab = labels.reshape(-1).long()
        own = (lab >= sh.lo) & (lab < sh.hi)
        lab_local = torch.where(own, lab - sh.lo, torch.full_like(lab, -1))  # -1: another shard owns it
        t16 = sh._tables(sh.ent_local, "local")
        loss_loc, lse_loc = sh.backend.ce_emb_fwd(t16, direction, rows, rel_rows, lab_local)
        true = torch.where(own, lse_loc - loss_loc, torch.zeros_like(lse_loc))  # the owner's label score
        if sh.collectives:
            allse = torch.empty(sh.world * lse_loc.numel(), dtype=lse_loc.dtype, device=lse_loc.device)
            dist.all_gather_into_tensor(allse, lse_loc.contiguous(), group=sh.group)
            lse = torch.logsumexp(allse.view(sh.world, -1), dim=0)
            dist.all_reduce(true, op=dist.ReduceOp.SUM, group=sh.group)
        else:
            lse = lse_loc
        ctx.sh, ctx.direction = sh, direction
        ctx.meta = (ids, p, lab_local, ent_master.shape, rel_master.shape)
        ctx.save_for_backward(rows, rel_rows, lse)
        return lse - true

    @staticmethod
    def backward(ctx, g_rows):
        sh = ctx.sh
        ids, p, lab_local, ent_shape, rel_shape = ctx.meta
        rows, rel_rows, lse = ctx.saved_tensors
        t16 = sh._tables(sh.ent_local, "local")
        g_a, g_p, g_t = sh.backend.ce_emb_bwd(t16, ctx.direction, rows, rel_rows, lab_local, lse,
                                              g_rows=g_rows.contiguous())
        d = g_a.shape[1]
        both = torch.cat([g_a, g_p], dim=1)  # this shard's part of the query-row gradients
        sh._allreduce(both)
        g_a, g_p = both[:, :d], both[:, d:]
        gid = ids.reshape(-1).long()
        own = ((gid >= sh.lo) & (gid < sh.hi)).to(g_a.dtype).unsqueeze(1)
        local = (gid - sh.lo).clamp_(0, max(sh.hi - sh.lo - 1, 0))
        ge = g_t  # [E_g, d], fresh: the gradient of this shard's rows as targets ...
        if sh.hi > sh.lo:
            ge.index_add_(0, local, g_a * own)  # ... plus the query rows it owns (others add zeros)
        gr = torch.zeros(rel_shape, dtype=torch.float32, device=g_p.device)
        gr.index_add_(0, p.reshape(-1).long(), g_p.contiguous())  # the same on every rank
        return None, None, ge.to(torch.float32).view(ent_shape), gr, None, None, None


class ShardedEntityTable:
    def __init__(self, scorer: str, ent_local: torch.Tensor, rel: torch.Tensor, num_entities: int,
                 l_norm: float = 1.0, group=None, backend=None, force_collectives: bool = False):
        self.scorer, self.l_norm = scorer, float(l_norm)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # With one rank every exchange step is an identity and is skipped -- unless `force_collectives` (or
        # KGE_SHARDED_FORCE_COLLECTIVES=1): then a one-rank process group still runs every all-gather /
        # all-reduce, so that a single-GPU box exercises the RCCL calls of the N > 1 path (tests, bench).
        force = force_collectives or os.environ.get("KGE_SHARDED_FORCE_COLLECTIVES") == "1"
        self.collectives = self.world > 1 or (force and dist.is_initialized())
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
        self._bufs, self._tcache = {}, {}
        self._lane = 0  # exchange buffers are per lane (ShardedScoreLanes: several batches in flight)
        # rank_batch_multi counts inside the scoring kernel where the backend offers it (no score slabs);
        # False / KGE_EVAL_TWO_STEP=1: score slabs + rank_counts_multi
        self.fused_rank = os.environ.get("KGE_EVAL_TWO_STEP", "0") != "1"

    @staticmethod
    def partition(num_entities: int, world: int, rank: int):
        shard = (num_entities + world - 1) // world
        lo = min(rank * shard, num_entities)
        return lo, min(lo + shard, num_entities)

    # ---- exchange steps -------------------------------------------------------------------
    def _allreduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.collectives:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def _tables(self, ent, key):
        """backend.Tables over `ent` + the replicated relation table, cached per buffer."""
        key = (self._lane, key)
        hit = self._tcache.get(key)
        if hit is None or hit[0] != ent.data_ptr():
            hit = (ent.data_ptr(), self.backend.Tables(self.scorer, ent, self.rel, self.l_norm))
            self._tcache[key] = hit
        return hit[1]

    def _buffer(self, key, shape, like):
        key = (self._lane, key)
        b = self._bufs.get(key)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != like.dtype:
            b = self._bufs[key] = torch.empty(shape, dtype=like.dtype, device=like.device)
        return b

    def exchange_rows(self, ids, rel_ids: Optional[torch.Tensor] = None):
        """Exchange step 1.  `ids` = list of k index vectors [n] of GLOBAL entity ids (the s and
        the o column of a batch): returns ([k*n, d] entity rows, vector after vector, and the
        [n, d_r] relation rows of `rel_ids` or None).  Two gather launches + one all-gather;
        nothing here waits for the device."""
        k, n = len(ids), ids[0].numel()
        d = self.ent_local.shape[1]
        if hasattr(self.backend, "shard_gather") and k <= 2 and self.hi > self.lo:
            # the id arithmetic inside the two gather kernels (kge_shard_gather / kge_shard_pick): three calls per
            # exchange instead of ten torch ops -- the step was bound by the HOST issuing them (90 us of Python per
            # 62 us of device work at the FB15k-237 shard shape)
            send = self._buffer(("send", k), (k * n, d), self.ent_local)
            rel_rows = None if rel_ids is None else self._buffer("rel", (n, self.rel.shape[1]), self.rel)
            self.backend.shard_gather(self._tables(self.ent_local, "local"), self.lo, ids, rel_ids, send, rel_rows)
            if not self.collectives:
                return send, rel_rows
            gath = self._buffer(("gath", k), (self.world * k * n, d), self.ent_local)
            dist.all_gather_into_tensor(gath.view(-1), send.view(-1), group=self.group)
            rows = self._buffer(("rows", k), (k * n, d), self.ent_local)
            self.backend.shard_pick(gath, self.shard, self.world, ids, rows)
            return rows, rel_rows
        gid = torch.cat([x.reshape(-1).long() for x in ids])
        local = (gid - self.lo).clamp_(0, max(self.hi - self.lo - 1, 0))  # not owned: any local row
        send = self._buffer(("send", k), (k * n, d), self.ent_local)
        rel_rows = None if rel_ids is None else self._buffer("rel", (n, self.rel.shape[1]), self.rel)
        if self.hi > self.lo:
            self.backend.embed(self._tables(self.ent_local, "local"), local, rel_ids, send, rel_rows)
        else:
            # a rank without rows (E = 9 over 4 ranks: rank 3 owns [9, 9)): nothing to gather from -- its block of
            # the all-gather is never picked (no id has this owner); zeros, and the relation rows on their own
            send.zero_()
            if rel_ids is not None:
                rel_rows.copy_(self.rel[rel_ids.reshape(-1).long()])
        if not self.collectives:
            return send, rel_rows
        gath = self._buffer(("gath", k), (self.world * k * n, d), self.ent_local)
        dist.all_gather_into_tensor(gath.view(-1), send.view(-1), group=self.group)
        owner = torch.div(gid, self.shard, rounding_mode="floor")
        pick = owner * (k * n) + torch.arange(k * n, device=gid.device)
        rows = self._buffer(("rows", k), (k * n, d), self.ent_local)
        self.backend.embed(self._tables(gath, ("gath", k)), pick, None, rows, None)
        return rows, rel_rows

    def gather_entity_rows(self, idx: torch.Tensor) -> torch.Tensor:
        """[n, d] rows of the GLOBAL entity table for global ids `idx`."""
        return self.exchange_rows([idx])[0].clone()

    # ---- scoring: local slabs, no collective ---------------------------------------------
    def score_sp(self, s: torch.Tensor, p: torch.Tensor):
        """[n, E_g]: scores of (s_i, p_i, ·) against this rank's entities."""
        rows, rel_rows = self.exchange_rows([s], p)
        return self.backend.score_emb(self.scorer, rows, rel_rows, self.ent_local, "sp_", self.l_norm)

    def score_po(self, p: torch.Tensor, o: torch.Tensor):
        rows, rel_rows = self.exchange_rows([o], p)
        return self.backend.score_emb(self.scorer, self.ent_local, rel_rows, rows, "_po", self.l_norm)

    # a score block beyond this many bytes no longer sits in the 256 MB Infinity Cache while it is
    # written: rows are then 128-byte aligned (padded pitch) and each direction gets its own launch
    BIG_SLAB_BYTES = 96 << 20

    def score_sp_po_blocks(self, s: torch.Tensor, p: torch.Tensor, o: torch.Tensor):
        """(score_sp slab, score_po slab), each [n, E_g] (row pitch >= E_g): ONE exchange for the s
        and the o rows, then one two-sided launch on the shard (backends with score_emb_sp_po; the
        blocks are the halves of its [n, 2 E_g] output) or, for slabs that outgrow the Infinity Cache,
        one launch per direction into matrices with a 128-byte-aligned row pitch."""
        n, m = s.numel(), self.hi - self.lo
        rows, rel_rows = self.exchange_rows([s, o], p)
        s_rows, o_rows = rows[:n], rows[n:]
        big = n * m * 4 > self.BIG_SLAB_BYTES
        if hasattr(self.backend, "score_emb_sp_po") and not big:
            both = self.backend.score_emb_sp_po(self.scorer, s_rows, rel_rows, o_rows, self.ent_local, self.l_norm)
            return both[:, :m], both[:, m:]
        kw = {"pad_pitch": True} if big else {}
        return (self.backend.score_emb(self.scorer, s_rows, rel_rows, self.ent_local, "sp_", self.l_norm, **kw),
                self.backend.score_emb(self.scorer, self.ent_local, rel_rows, o_rows, "_po", self.l_norm, **kw))

    def score_sp_po(self, s: torch.Tensor, p: torch.Tensor, o: torch.Tensor):
        """[n, 2 E_g]: the two slabs of score_sp_po_blocks side by side (KgeModel.score_sp_po's layout)."""
        sp, po = self.score_sp_po_blocks(s, p, o)
        if sp.data_ptr() + sp.shape[1] * 4 == po.data_ptr() and sp.stride(0) == 2 * sp.shape[1]:
            return torch.as_strided(sp, (sp.shape[0], 2 * sp.shape[1]), (sp.stride(0), 1))
        return torch.cat([sp, po], dim=1)

    # ---- training: 1vsAll cross entropy over the entities of all shards -----------------------------
    def ce_loss(self, direction: str, ids: torch.Tensor, p: torch.Tensor, labels: torch.Tensor,
                ent_master: Optional[torch.Tensor] = None, rel_master: Optional[torch.Tensor] = None):
        """[n] cross entropy of score_sp(ids, p) ("sp": ids = subjects, labels = true objects) or
        score_po(p, ids) ("po": ids = objects, labels = true subjects) over ALL entities; sum / batch
        size = the reference's 1vsAll loss (train_1vsAll.py:64-81, loss.py:192-207).  Differentiable
        w.r.t. `ent_master` (this rank's shard of the entity parameters, [E_g, d]) and `rel_master` (the
        replicated relation parameters): float32 masters whose bf16 scoring copies are this object's
        tables (refresh them with `refresh_tables` after an optimizer step), or omitted when the tables
        themselves are the parameters."""
        ent_master = self.ent_local if ent_master is None else ent_master
        rel_master = self.rel if rel_master is None else rel_master
        return _ShardedCE.apply(self, direction, ent_master, rel_master, ids, p, labels)

    @torch.no_grad()
    def refresh_tables(self, ent_master: torch.Tensor, rel_master: torch.Tensor):
        """Re-cast the float32 masters into this object's scoring tables IN PLACE (same storage: cached
        kernel descriptors and exchange buffers stay valid)."""
        self.ent_local.copy_(ent_master)
        self.rel.copy_(rel_master)

    def true_scores(self, slab: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """Score of each row's true entity, taken from the owner's slab (exchange step 2)."""
        target = target.long()
        own = (target >= self.lo) & (target < self.hi)
        col = (target - self.lo).clamp_(0, max(slab.shape[1] - 1, 0))
        t = torch.where(own, slab.gather(1, col.view(-1, 1)).view(-1), torch.zeros((), dtype=slab.dtype,
                                                                                  device=slab.device))
        return self._allreduce(t.float())

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
        sp, po = self.score_sp_po_blocks(s, p, o)
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
        if (self.fused_rank and K <= 2 and hasattr(self.backend, "score_rank_emb_sp_po")
                and hasattr(self.backend, "score_emb_sp_po")):
            counts = self._rank_batch_fused(s, p, o, filters_o, filters_s, atol, rtol)
            if counts is not None:
                return counts
        sp, po = self.score_sp_po_blocks(s, p, o)
        o_true = self.true_scores(sp, o)
        s_true = self.true_scores(po, s)
        counts = torch.zeros(2, 2, K + 1, n, dtype=torch.int64, device=sp.device)
        self.backend.rank_counts_multi(sp, o_true, filters_o, self.lo, o.long().contiguous(), atol, rtol,
                                       counts[0, 0], counts[0, 1])
        self.backend.rank_counts_multi(po, s_true, filters_s, self.lo, s.long().contiguous(), atol, rtol,
                                       counts[1, 0], counts[1, 1])
        return self._allreduce(counts)

    def _rank_batch_fused(self, s, p, o, filters_o, filters_s, atol, rtol):
        """rank_batch_multi without the score slabs: the counts come out of the scoring kernel
        (kge_score_rank_emb_sp_po over this rank's shard).  The true scores need no exchange of their own: the
        exchanged rows of s and o are on every rank, so each rank scores every query against its own target row
        (one two-sided launch on 2n target rows, the diagonals kept) -- the bits the owner's slab would hold.
        None: the backend declines (tables other than bf16 ComplEx / DistMult, dim 256 / 512)."""
        n, K = s.numel(), len(filters_o)
        rows, rel_rows = self.exchange_rows([o, s], p)
        o_rows, s_rows = rows[:n], rows[n:]
        both = self.backend.score_emb_sp_po(self.scorer, s_rows, rel_rows, o_rows, rows, self.l_norm)  # [n, 4n]
        o_true = both.as_strided((n,), (4 * n + 1,)).contiguous()
        s_true = both.as_strided((n,), (4 * n + 1,), 3 * n).contiguous()
        counts = torch.zeros(2, 2, K + 1, n, dtype=torch.int64, device=rows.device)
        ok = self.backend.score_rank_emb_sp_po(self.scorer, s_rows, rel_rows, o_rows, s, o, self.ent_local, self.lo,
                                               o_true, s_true, filters_o, filters_s, atol, rtol, counts[0, 0],
                                               counts[0, 1], counts[1, 0], counts[1, 1], self.l_norm)
        if not ok:  # the same answer on every rank (dtype / scorer / dim): two steps from now on
            self.fused_rank = False
            return None
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
        if self.collectives:
            vs = [torch.empty_like(v) for _ in range(self.world)]
            is_ = [torch.empty_like(i) for _ in range(self.world)]
            dist.all_gather(vs, v, group=self.group)
            dist.all_gather(is_, i, group=self.group)
            v, i = torch.cat(vs, 1), torch.cat(is_, 1)
        tv, ti = torch.topk(v, k, dim=1)
        return tv, torch.gather(i, 1, ti)


class ShardedScoreLanes:
    """Several batches in flight over one ShardedEntityTable: batch k runs on HIP stream k % L -- its row exchange
    (gather launch, RCCL all-gather, pick launch) and its scoring launch, in that order on that stream -- so the
    exchange of batch k + 1 (39 of 62 us of a step at the FB15k-237 shard shape, DESIGN.md 6: collective latency, the
    compute units idle) runs under the scoring launch of batch k, and the scoring launch of batch k + 1 starts in
    the holes batch k's leaves (ScorePipeline(streams=L) in kge_amd/engine.py has the single-GPU measurements).

    Every rank issues the same batches in the same order, so the collectives of the lanes reach the communicator in
    one order on all ranks (torch.distributed runs them on the process group's own stream, ordered against the
    lane's stream by events).  Exchange buffers are per lane.

    graph (default: on without collectives and for a one-rank RCCL group, opt-in -- graph=True / KGE_SHARDED_GRAPH=1 --
    with more ranks; KGE_SHARDED_GRAPH=0 turns it off): a lane's step -- two gather
    launches, the all-gather, the scoring launch -- is captured into a hipGraph on first use (per batch shape) and
    replayed from then on with the batch's ids copied into static index vectors: the step issued from Python is
    bound by the HOST (55 us of calls for ~45 us of device work at the FB15k-237 shard shape, one rank), a copy and
    a replay are ~15 us.  A capture that fails (a backend that stages collectives through the host) turns the
    feature off and the step is issued call by call.

    The score slabs of batch k belong to lane k % L: fresh tensors of the lane's stream (call by call) or the
    lane's static outputs (graph), valid after join() and until that lane's NEXT batch is issued -- fork() after
    reading them, as with engine.ScorePipeline:

        lanes = ShardedScoreLanes(table, 2)
        for k, (s, p, o) in enumerate(batches):
            pending.append(lanes.score_sp_po_blocks(s, p, o))
            if len(pending) == 2: lanes.join(); consume(pending); pending = []; lanes.fork()
    """

    def __init__(self, table: ShardedEntityTable, lanes: int = 2, graph: Optional[bool] = None):
        self.table = table
        self.L = max(1, int(lanes))
        dev = table.ent_local.device
        self.cuda = dev.type == "cuda"
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(self.L)] if self.cuda and self.L > 1 else None
        self.k = 0
        self._out = []
        if graph is None:
            # default: ON for every world size (KGE_SHARDED_GRAPH=0 turns it off) behind a self-check: the first
            # capture of a step shape is replayed once and compared bit for bit with the step issued call by call,
            # every rank votes (one all-reduce), and a capture that throws, differs or is voted down anywhere turns
            # the feature off on all ranks -- loudly (warnings.warn + `graph_error`) --, the step going call by call
            # from then on.  (The capture of a multi-GPU RCCL all-gather has run on ONE rank only in the build loop.)
            want = os.environ.get("KGE_SHARDED_GRAPH")
            graph = (want != "0") if want is not None else True
            if graph and table.collectives:
                try:
                    graph = dist.get_backend(table.group) == "nccl"
                except Exception:  # pragma: no cover
                    graph = False
        self.use_graph = bool(graph) and self.cuda
        self._graphs = [dict() for _ in range(self.L)]
        self._cap_stream = None
        self.graph_replays = 0
        self.graph_error = None
        self.graph_checked = 0  # captures that passed the self-check

    def fork(self):
        """The lanes wait for torch's current stream (producers of the batches; readers of earlier results)."""
        if self.streams is not None:
            ev = torch.cuda.Event()
            ev.record()
            for st in self.streams:
                st.wait_event(ev)

    def join(self):
        """Torch's current stream waits for every lane; the slabs returned so far now belong to it."""
        if self.streams is not None:
            cur = torch.cuda.current_stream(self.table.ent_local.device)
            for st in self.streams:
                ev = torch.cuda.Event()
                ev.record(st)
                cur.wait_event(ev)
            for t in self._out:
                t.record_stream(cur)
        self._out = []

    def _issue(self, lane, name, fn, args):
        """On the lane's stream (already current): call by call, or copy + replay of the lane's captured step."""
        if not self.use_graph:
            return fn(*args), False
        # (a capture holds the tables' addresses: tables re-allocated behind our back get a capture of their own --
        # refresh_tables copies in place and keeps them valid)
        tb = self.table
        key = (name, tb.ent_local.data_ptr(), tb.rel.data_ptr()) + tuple((tuple(a.shape), a.dtype) for a in args)
        ent = self._graphs[lane].get(key)
        if ent is not None:
            for d, x in zip(ent["static"], args):
                d.copy_(x)
            ent["graph"].replay()
            self.graph_replays += 1
            return ent["res"], True
        static = [torch.empty(a.shape, dtype=a.dtype, device=a.device) for a in args]
        for d, x in zip(static, args):
            d.copy_(x)
        res = fn(*static)  # once call by call: per-stream scratch gets allocated, the communicator warmed up
        st = torch.cuda.current_stream(self.table.ent_local.device)
        if self.streams is None:  # a capture needs a stream of its own
            if self._cap_stream is None:
                self._cap_stream = torch.cuda.Stream(device=self.table.ent_local.device)
            st = self._cap_stream
        err = None
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                cap = fn(*static)
            # self-check: one replay against the call-by-call result on the same static inputs
            g.replay()
            torch.cuda.synchronize()
            flat = lambda r: [t for t in (r if isinstance(r, (tuple, list)) else (r,)) if torch.is_tensor(t)]
            same = all(torch.equal(a, b) for a, b in zip(flat(res), flat(cap))) and len(flat(res)) == len(flat(cap))
            if not same:
                err = "the replayed step differs from the step issued call by call"
        except Exception as exc:  # not capturable here
            err = f"{type(exc).__name__}: {exc}"
            torch.cuda.synchronize()
        if tb.collectives and tb.world > 1:  # every rank takes the same decision
            try:
                vote = torch.tensor([0 if err is None else 1], device=tb.ent_local.device, dtype=torch.int32)
                dist.all_reduce(vote, op=dist.ReduceOp.MAX, group=tb.group)
                if int(vote.item()) != 0 and err is None:
                    err = "another rank's capture failed its self-check"
            except Exception as exc:  # pragma: no cover
                err = err or f"vote failed: {type(exc).__name__}: {exc}"
        if err is None:
            self._graphs[lane][key] = {"graph": g, "static": static, "res": cap}
            self.graph_checked += 1
        else:
            self.use_graph = False
            self.graph_error = err
            import warnings
            warnings.warn(f"kge_amd.sharded: hipGraph capture of the sharded step disabled ({err}); the step is "
                          f"issued call by call", RuntimeWarning)
        return res, False

    def _run(self, name, fn, *args):
        lane = self.k % self.L
        self.k += 1
        tb = self.table
        if not self.cuda:
            tb._lane = lane
            try:
                return fn(*args)
            finally:
                tb._lane = 0
        prev = torch.cuda.current_stream(tb.ent_local.device)
        if self.streams is not None:
            torch.cuda.set_stream(self.streams[lane])  # (the context manager costs ~3x this pair)
        tb._lane = lane
        try:
            res, static_out = self._issue(lane, name, fn, args)
        finally:
            tb._lane = 0
            if self.streams is not None:
                torch.cuda.set_stream(prev)
        if self.streams is not None and not static_out:
            self._out.extend(r for r in (res if isinstance(res, tuple) else (res,)) if torch.is_tensor(r))
        return res

    def score_sp_po_blocks(self, s, p=None, o=None):
        """(s, p, o) index vectors, or the batch as ONE [n, 3] triples tensor (one copy into the lane's static ids)."""
        if p is None and o is None and s.dim() == 2:
            tb = self.table
            return self._run("sp_po3", lambda t: tb.score_sp_po_blocks(t[:, 0], t[:, 1], t[:, 2]), s)
        return self._run("sp_po", self.table.score_sp_po_blocks, s, p, o)

    def score_sp(self, s, p):
        return self._run("sp", self.table.score_sp, s, p)

    def score_po(self, p, o):
        return self._run("po", self.table.score_po, p, o)
