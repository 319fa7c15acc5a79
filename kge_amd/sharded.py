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
        ge, gr = _merge_query_grads(sh, ids, p, g_a, g_p, g_t, ent_shape, rel_shape)
        return None, None, ge, gr, None, None, None


def _merge_query_grads(sh, ids, p, g_a, g_p, g_t, ent_shape, rel_shape):
    """The tail of every sharded loss backward: this shard's part of the query-row gradients summed over the shards
    (ONE all-reduce of [n, d + d_r] floats), the owners scatter-add them into the gradient of their own rows (g_t:
    fresh, never leaves the rank); the relation gradient comes out identical on every rank."""
    d = g_a.shape[1]
    both = torch.cat([g_a, g_p], dim=1)
    sh._allreduce(both)
    g_a, g_p = both[:, :d], both[:, d:]
    gid = ids.reshape(-1).long()
    own = ((gid >= sh.lo) & (gid < sh.hi)).to(g_a.dtype).unsqueeze(1)
    local = (gid - sh.lo).clamp_(0, max(sh.hi - sh.lo - 1, 0))
    ge = g_t
    if sh.hi > sh.lo:
        ge.index_add_(0, local, g_a * own)
    gr = torch.zeros(rel_shape, dtype=torch.float32, device=g_p.device)
    gr.index_add_(0, p.reshape(-1).long(), g_p.contiguous())
    return ge.to(torch.float32).view(ent_shape), gr


def _kl_label_terms(rowptr, eps: float, num_entities: int):
    """Per-row constants of the KvsAll KL loss (train_KvsAll.py:244-294, loss.py:208-213) -> (k, has, w, bias, const):
    the loss row is  lse_i - w_i sum_{labels} x_ij - bias_i sum_{all j} x_ij + const_i  on rows with `has`, 0 elsewhere.
    Without label smoothing: w = 1 / k, bias = None, const = -log k, rows without labels contribute nothing.  With
    smoothing eps (train_KvsAll.py:260-266: labels = (1 - eps) multi_hot + 1 / E, then normalised): the label row is a_i
    on the k_i labels and b_i elsewhere, Z_i = (1 - eps) k_i + 1, a_i = (1 - eps + 1 / E) / Z_i, b_i = (1 / E) / Z_i:
    w = a - b, bias = b, const = k a log a + (E - k) b log b (kge_amd.model.kl_fused has the unsharded form)."""
    k = (rowptr[1:] - rowptr[:-1]).to(torch.float32)
    if eps == 0.0:
        has = k > 0
        w = torch.where(has, 1.0 / k.clamp(min=1.0), torch.zeros_like(k))
        return k, has, w, None, -torch.log(k.clamp(min=1.0))
    E = float(num_entities)
    Z = (1.0 - eps) * k + 1.0
    a_w, b_w = (1.0 - eps + 1.0 / E) / Z, (1.0 / E) / Z
    const = k * a_w * torch.log(a_w) + (E - k) * b_w * torch.log(b_w)
    return k, torch.ones_like(k, dtype=torch.bool), (a_w - b_w).contiguous(), b_w.contiguous(), const


@torch.no_grad()
def _sum_of_shard_scores(sh, t16, direction, a_rows, p_rows):
    """[n]: sum_j score(i, j) over THIS shard's entities j -- ComplEx and DistMult are linear in the target row, so it is
    one score of every query against the shard's column sum (the label-smoothing term's value; its gradient is taken
    inside the gradient kernel: label_bias)."""
    colsum = t16.ent.float().sum(dim=0, keepdim=True)
    a32, p32 = a_rows.float(), p_rows.float()
    if direction == "sp":
        return sh.backend.score_emb(sh.scorer, a32, p32, colsum, "sp_", sh.l_norm).reshape(-1).to(torch.float32)
    return sh.backend.score_emb(sh.scorer, colsum, p32, a32, "_po", sh.l_norm).reshape(-1).to(torch.float32)


class _ShardedKL(torch.autograd.Function):
    """Per-row KL divergence of softmax(score(i, .)) over the entities of ALL shards from the row's normalised
    multi-hot labels (TrainingJobKvsAll with train.loss: kl, kge/job/train_KvsAll.py:244-294, kge/util/loss.py:208-213):
    loss_i = lse_i - (1 / k_i) sum_{j in labels_i} score(i, j) - log k_i  (0 for a row without labels).  The label CSR
    holds GLOBAL entity ids and is the same on every rank: each shard's kernel takes the labels it owns."""

    @staticmethod
    def forward(ctx, sh, direction, ent_master, rel_master, ids, p, rowptr, col, eps=0.0):
        rows, rel_rows = sh.exchange_rows([ids], p)
        rows, rel_rows = rows.clone(), rel_rows.clone()
        k, has, w, bias, const = _kl_label_terms(rowptr, float(eps), sh.E)
        t16 = sh._tables(sh.ent_local, "local")
        loss_loc, lse_loc = sh.backend.kl_emb_fwd(t16, direction, rows, rel_rows, rowptr, col, sh.lo, w)
        # w_i * (sum of the label scores inside this shard).  A rank whose shard is EMPTY (E = 9 over 4 ranks) has
        # lse = -inf and loss = -inf: the difference would be NaN and the all-reduce below would spread it.
        lab = torch.where(torch.isfinite(lse_loc), lse_loc - loss_loc, torch.zeros_like(lse_loc))
        if bias is not None:  # label smoothing: + bias_i x (this shard's sum of ALL scores of row i), one all-reduce for both
            lab = lab + bias * _sum_of_shard_scores(sh, t16, direction, rows, rel_rows)
        if sh.collectives:
            allse = torch.empty(sh.world * lse_loc.numel(), dtype=lse_loc.dtype, device=lse_loc.device)
            dist.all_gather_into_tensor(allse, lse_loc.contiguous(), group=sh.group)
            lse = torch.logsumexp(allse.view(sh.world, -1), dim=0)
            dist.all_reduce(lab, op=dist.ReduceOp.SUM, group=sh.group)
        else:
            lse = lse_loc
        ctx.sh, ctx.direction = sh, direction
        ctx.meta = (ids, p, rowptr, col, ent_master.shape, rel_master.shape)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(rows, rel_rows, lse, w, has, bias if bias is not None else w)
        return torch.where(has, lse - lab + const, torch.zeros_like(lse))

    @staticmethod
    def backward(ctx, g_rows):
        sh = ctx.sh
        ids, p, rowptr, col, ent_shape, rel_shape = ctx.meta
        rows, rel_rows, lse, w, has, bias = ctx.saved_tensors
        g = torch.where(has, g_rows, torch.zeros_like(g_rows)).contiguous()  # rows without labels: no gradient
        t16 = sh._tables(sh.ent_local, "local")
        g_a, g_p, g_t = sh.backend.kl_emb_bwd(t16, ctx.direction, rows, rel_rows, rowptr, col, sh.lo, w, lse, g_rows=g,
                                              label_bias=bias if ctx.has_bias else None)
        ge, gr = _merge_query_grads(sh, ids, p, g_a, g_p, g_t, ent_shape, rel_shape)
        return None, None, ge, gr, None, None, None, None, None


class _ShardedBCE(torch.autograd.Function):
    """Per-row sum over the entities of ALL shards of BCEWithLogits(score(i, j) + offset, y_ij), y = 1 on the row's
    labels (train.loss: bce; kge/util/loss.py:137-159).  Additive over the shards: one all-reduce of n floats, no
    statistic to merge."""

    @staticmethod
    def forward(ctx, sh, direction, ent_master, rel_master, ids, p, rowptr, col, offset):
        rows, rel_rows = sh.exchange_rows([ids], p)
        rows, rel_rows = rows.clone(), rel_rows.clone()
        t16 = sh._tables(sh.ent_local, "local")
        loss = sh.backend.bce_emb_fwd(t16, direction, rows, rel_rows, rowptr, col, sh.lo, offset)
        loss = sh._allreduce(loss.clone())
        ctx.sh, ctx.direction, ctx.offset = sh, direction, float(offset)
        ctx.meta = (ids, p, rowptr, col, ent_master.shape, rel_master.shape)
        ctx.save_for_backward(rows, rel_rows)
        return loss

    @staticmethod
    def backward(ctx, g_rows):
        sh = ctx.sh
        ids, p, rowptr, col, ent_shape, rel_shape = ctx.meta
        rows, rel_rows = ctx.saved_tensors
        t16 = sh._tables(sh.ent_local, "local")
        g_a, g_p, g_t = sh.backend.bce_emb_bwd(t16, ctx.direction, rows, rel_rows, rowptr, col, sh.lo, ctx.offset,
                                               g_rows=g_rows.contiguous())
        ge, gr = _merge_query_grads(sh, ids, p, g_a, g_p, g_t, ent_shape, rel_shape)
        return None, None, ge, gr, None, None, None, None, None


# ---- embedder dropout on the sharded table (round 6) ----------------------------------------------------------------
# LookupEmbedder._postprocess (kge/model/embedder/lookup_embedder.py:64-69, 102-105) applies torch.nn.Dropout to what
# embed() / embed_all() return: in a training step score_sp(s, p) scores dropout(E[s]) (x) dropout(R[p]) against
# dropout(E) -- three independent masks (kge_model.py:682-725: s rows, p rows, all entities for sp_; all entities, o
# rows, p rows for _po).  Over a sharded table: the batch's query rows are brought to every rank in FLOAT32 from their
# owners' masters (_OwnerRows: one all-reduce of [n, d], x + 0 is exact), masked identically on every rank (the ranks
# draw from generators seeded alike, sharded_job._seed_epoch), every rank masks ITS rows of the table, and the dense-row
# loss kernels (kge_ce_emb_* / kge_kl_weighted_emb_* / kge_bce_emb_*) run on the masked rows (_ShardedDense: the
# per-shard step and merges of _ShardedCE / _ShardedKL / _ShardedBCE with the rows handed in); autograd carries the
# gradients back through the masks, sums the query rows' over the shards and scatters them on their owners.
def _drop(x, p, mask=None, generator=None):
    """torch.nn.functional.dropout(x, p, training=True), the mask optionally handed in (0 / 1, same shape) or drawn
    from `generator` (the table's per-rank generator)."""
    if p <= 0.0:
        return x
    if mask is None:
        if generator is None:
            return torch.nn.functional.dropout(x, p, True)
        mask = torch.empty_like(x, dtype=torch.float32).bernoulli_(1.0 - p, generator=generator)
    return x * (mask.to(x.dtype) * (1.0 / (1.0 - p)))


class _OwnerRows(torch.autograd.Function):
    """[n, d] float32 rows of the GLOBAL ids `ids` from the sharded master, on every rank."""

    @staticmethod
    def forward(ctx, sh, ent_master, ids):
        gid = ids.reshape(-1).long()
        own = ((gid >= sh.lo) & (gid < sh.hi)).to(ent_master.dtype).unsqueeze(1)
        local = (gid - sh.lo).clamp_(0, max(sh.hi - sh.lo - 1, 0))
        rows = ent_master.detach()[local] * own if sh.hi > sh.lo else torch.zeros(
            gid.numel(), ent_master.shape[1], dtype=ent_master.dtype, device=ent_master.device)
        sh._allreduce(rows)  # every id has one owner: value + zeros
        ctx.sh, ctx.shape = sh, ent_master.shape
        ctx.save_for_backward(local, own)
        return rows

    @staticmethod
    def backward(ctx, g):
        sh = ctx.sh
        local, own = ctx.saved_tensors
        g = g.contiguous().clone()
        sh._allreduce(g)  # every shard's part of the query rows' gradient
        ge = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        if sh.hi > sh.lo:
            ge.index_add_(0, local, g * own)
        return None, ge, None


class _SumOverShards(torch.autograd.Function):
    """Identity whose gradient is summed over the ranks (the relation rows' gradient: every shard contributes)."""

    @staticmethod
    def forward(ctx, sh, x):
        ctx.sh = sh
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        ctx.sh._allreduce(g)
        return None, g


class _ShardedDense(torch.autograd.Function):
    """The per-shard step of _ShardedCE / _ShardedKL / _ShardedBCE on rows that are handed in: `table` [E_g, d] (this
    rank's rows), `a_rows` [n, d], `p_rows` [n, d_r] -- float32, rounded to the scoring dtype inside; gradients come
    back in float32 w.r.t. all three (this SHARD's part of them: the callers' _OwnerRows / _SumOverShards sum)."""

    @staticmethod
    def forward(ctx, sh, kind, direction, table, a_rows, p_rows, extra):
        sd = sh.ent_local.dtype
        t16 = sh.backend.Tables(sh.scorer, table.detach().to(sd).contiguous(), sh.rel, sh.l_norm)
        a16, p16 = a_rows.detach().to(sd).contiguous(), p_rows.detach().to(sd).contiguous()
        ctx.sh, ctx.kind, ctx.direction, ctx.t16 = sh, kind, direction, t16

        def merge(lse_loc):
            if not sh.collectives:
                return lse_loc
            allse = torch.empty(sh.world * lse_loc.numel(), dtype=lse_loc.dtype, device=lse_loc.device)
            dist.all_gather_into_tensor(allse, lse_loc.contiguous(), group=sh.group)
            return torch.logsumexp(allse.view(sh.world, -1), dim=0)
        if kind == "ce":
            lab = extra[0].reshape(-1).long()
            own = (lab >= sh.lo) & (lab < sh.hi)
            lab_local = torch.where(own, lab - sh.lo, torch.full_like(lab, -1))
            loss_loc, lse_loc = sh.backend.ce_emb_fwd(t16, direction, a16, p16, lab_local)
            true = torch.where(own, lse_loc - loss_loc, torch.zeros_like(lse_loc))
            lse = merge(lse_loc)
            sh._allreduce(true)
            ctx.extra = (lab_local,)
            ctx.save_for_backward(a16, p16, lse)
            return lse - true
        rowptr, col = extra[0], extra[1]
        if kind == "kl":
            eps = float(extra[2]) if len(extra) > 2 else 0.0
            k, has, w, bias, const = _kl_label_terms(rowptr, eps, sh.E)
            loss_loc, lse_loc = sh.backend.kl_emb_fwd(t16, direction, a16, p16, rowptr, col, sh.lo, w)
            lab = torch.where(torch.isfinite(lse_loc), lse_loc - loss_loc, torch.zeros_like(lse_loc))
            if bias is not None:
                lab = lab + bias * _sum_of_shard_scores(sh, t16, direction, a16, p16)
            lse = merge(lse_loc)
            sh._allreduce(lab)
            ctx.extra = (rowptr, col)
            ctx.has_bias = bias is not None
            ctx.save_for_backward(a16, p16, lse, w, has, bias if bias is not None else w)
            return torch.where(has, lse - lab + const, torch.zeros_like(lse))
        offset = float(extra[2])
        loss = sh.backend.bce_emb_fwd(t16, direction, a16, p16, rowptr, col, sh.lo, offset)
        loss = sh._allreduce(loss.clone())
        ctx.extra = (rowptr, col, offset)
        ctx.save_for_backward(a16, p16)
        return loss

    @staticmethod
    def backward(ctx, g_rows):
        sh, t16 = ctx.sh, ctx.t16
        if ctx.kind == "ce":
            a16, p16, lse = ctx.saved_tensors
            g_a, g_p, g_t = sh.backend.ce_emb_bwd(t16, ctx.direction, a16, p16, ctx.extra[0], lse, g_rows=g_rows.contiguous())
        elif ctx.kind == "kl":
            a16, p16, lse, w, has, bias = ctx.saved_tensors
            g = torch.where(has, g_rows, torch.zeros_like(g_rows)).contiguous()
            g_a, g_p, g_t = sh.backend.kl_emb_bwd(t16, ctx.direction, a16, p16, ctx.extra[0], ctx.extra[1], sh.lo, w, lse, g_rows=g,
                                                  label_bias=bias if ctx.has_bias else None)
        else:
            a16, p16 = ctx.saved_tensors
            g_a, g_p, g_t = sh.backend.bce_emb_bwd(t16, ctx.direction, a16, p16, ctx.extra[0], ctx.extra[1], sh.lo,
                                                   ctx.extra[2], g_rows=g_rows.contiguous())
        return None, None, None, g_t.to(torch.float32), g_a.to(torch.float32), g_p.to(torch.float32), None


class _ShardedNeg(torch.autograd.Function):
    """Scores of a negative-sampling batch over the sharded table (TrainingJobNegativeSampling._process_subbatch,
    kge/job/train_negative_sampling.py:103-164, with BatchNegativeSample.score, kge/util/sampler.py:263-306): the n
    positives and, for one slot (0 = s, 2 = o), the [n, K] triples with that slot replaced by negatives (GLOBAL ids,
    the same on every rank).

    The rank's table has 2 n_max SLACK rows behind its shard (ShardedEntityTable.with_slack): the exchanged s and o
    rows of the batch are written there, so that the index-level kernels (kge_score_spo, kge_score_neg and their
    backward twins: fixed side in registers, only the corrupted rows stream) run unchanged on local row ids.  A rank
    scores the negatives it OWNS (the others: any local row, masked to 0) and one all-reduce of [n, K] floats (x + 0 is
    exact) gives every rank the slot's score block; the positives come out identical on every rank.  Backward: the
    corrupted rows' gradients stay on their owner; the slack rows' (= query rows') and the relation gradients are
    all-reduced, the owners scatter-add."""

    @staticmethod
    def forward(ctx, sh, ent_master, rel_master, s, p, o, slot, neg):
        n, K = neg.shape
        Eg = sh.hi - sh.lo
        ext = sh.ent_ext  # [E_g + slack, d]: the shard + slack rows, the master's own storage
        if 2 * n > ext.shape[0] - Eg:
            raise ValueError("kge_amd: batch larger than the slack rows of the sharded table (with_slack(n_max))")
        rows, _ = sh.exchange_rows([s, o], None)
        with torch.no_grad():
            ext[Eg:Eg + 2 * n].copy_(rows)
        ar = torch.arange(n, device=neg.device)
        si, oi = Eg + ar, Eg + n + ar
        T = sh._tables(ext, "ext")
        pos = sh.backend.score_spo(T, si, p, oi)
        own = (neg >= sh.lo) & (neg < sh.hi)
        local = (neg - sh.lo).clamp(0, max(Eg - 1, 0))
        sc = sh.backend.score_neg(T, si, p, oi, int(slot), local)
        sc = torch.where(own, sc, torch.zeros_like(sc))
        sc = sh._allreduce(sc)
        ctx.sh, ctx.slot = sh, int(slot)
        ctx.meta = (s, p, o, ent_master.shape, rel_master.shape)
        ctx.save_for_backward(si, oi, local, own, pos, sc)
        return pos, sc

    @staticmethod
    def backward(ctx, g_pos, g_neg):
        sh = ctx.sh
        s, p, o, ent_shape, rel_shape = ctx.meta
        si, oi, local, own, pos, sc = ctx.saved_tensors
        n = si.numel()
        Eg = sh.hi - sh.lo
        ext = sh.ent_ext
        T = sh._tables(ext, "ext")
        ge = torch.zeros(ext.shape, dtype=torch.float32, device=ext.device)
        gr = torch.zeros(rel_shape, dtype=torch.float32, device=ext.device)
        gneg = torch.where(own, g_neg, torch.zeros_like(g_neg)).contiguous()
        if not sh.backend.score_neg_bwd_accum(T, si, p, oi, ctx.slot, local, gneg, sc, ge, gr):
            raise RuntimeError("kge_amd: kge_score_neg_bwd_accum declined the sharded negative-sampling shape")
        if sh.rank == 0 or not sh.collectives:  # the positives are the same on every rank: one of them contributes
            sh.backend.score_spo_bwd_accum(T, si, p, oi, g_pos.contiguous(), pos, ge, gr)
        # query-row (slack) and relation gradients: summed over the shards, then scatter-added by the owners
        q = ge[Eg:Eg + 2 * n].clone()
        d = q.shape[1]
        pack = torch.cat([q.reshape(-1), gr.reshape(-1)])
        sh._allreduce(pack)
        q, gr = pack[:2 * n * d].view(2 * n, d), pack[2 * n * d:].view(rel_shape)
        gid = torch.cat([s.reshape(-1).long(), o.reshape(-1).long()])
        ownq = ((gid >= sh.lo) & (gid < sh.hi)).to(q.dtype).unsqueeze(1)
        loc = (gid - sh.lo).clamp_(0, max(Eg - 1, 0))
        out = ge[:Eg]
        if Eg > 0:
            out.index_add_(0, loc, q * ownq)
        return None, out.view(ent_shape), gr, None, None, None, None, None


class ShardedEntityTable:
    # measurement / test settings, class attributes (until round 6 the environment variables
    # KGE_SHARDED_FORCE_COLLECTIVES and KGE_EVAL_TWO_STEP): FORCE_COLLECTIVES = a one-rank process group still issues
    # every collective; TWO_STEP = rank_batch_multi never counts inside the scoring kernel
    FORCE_COLLECTIVES = False
    TWO_STEP = False

    def __init__(self, scorer: str, ent_local: torch.Tensor, rel: torch.Tensor, num_entities: int,
                 l_norm: float = 1.0, group=None, backend=None, force_collectives: bool = False):
        self.scorer, self.l_norm = scorer, float(l_norm)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # With one rank every exchange step is an identity and is skipped -- unless `force_collectives` (or the class
        # attribute FORCE_COLLECTIVES): then a one-rank process group still runs every all-gather /
        # all-reduce, so that a single-GPU box exercises the RCCL calls of the N > 1 path (tests, bench).
        force = force_collectives or self.FORCE_COLLECTIVES
        self.collectives = self.world > 1 or (force and dist.is_initialized())
        self.E = int(num_entities)
        self.shard = (self.E + self.world - 1) // self.world
        self.check_partition(self.E, self.world)
        self.lo = min(self.rank * self.shard, self.E)
        self.hi = min(self.lo + self.shard, self.E)
        if ent_local.shape[0] != self.hi - self.lo:
            raise ValueError(f"rank {self.rank} must hold rows [{self.lo},{self.hi}) of the entity table")
        self.ent_local, self.rel = ent_local, rel
        self.ent_ext = None  # negative sampling: the shard + slack rows (with_slack), set by the training job
        if backend is None:
            from . import engine as backend  # the HIP kernels; no CPU fallback
        self.backend = backend
        self._bufs, self._tcache = {}, {}
        self._lane = 0  # exchange buffers are per lane (ShardedScoreLanes: several batches in flight)
        # A reciprocal-relations model (kge/model/reciprocal_relations_model.py): the relation table holds 2 R rows and the
        # subject direction is an sp_ query with relation p + R -- score_po(p, o) = score_sp(o, p + R).  0 = a plain model.
        # Every "po" of this class is translated in ONE place (_recip); negative sampling is not offered.
        self.reciprocal_R = 0
        # rank_batch_multi counts inside the scoring kernel where the backend offers it (no score slabs);
        # False (TWO_STEP): score slabs + rank_counts_multi
        self.fused_rank = not self.TWO_STEP

    @staticmethod
    def check_partition(num_entities: int, world: int):
        """Every rank must own at least one row: the kernels refuse an empty table (KGE_ERR_INVALID_ARG), and a rank
        failing alone would leave the others waiting in the next collective.  The split is a function of
        (num_entities, world) only, so every rank raises here together."""
        shard = (num_entities + world - 1) // world
        if num_entities < 1 or (world - 1) * shard >= num_entities:
            raise ValueError(f"kge_amd: {num_entities} entities in shards of {shard} rows leave rank(s) from "
                             f"{(num_entities + shard - 1) // max(shard, 1)} of {world} without rows; use fewer ranks")

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

    def _recip(self, direction: str, p: torch.Tensor):
        """(direction, relation ids) as the kernels see them: a reciprocal model's "po" is "sp" with p + R."""
        if self.reciprocal_R and direction == "po":
            return "sp", p + self.reciprocal_R
        return direction, p

    # ---- scoring: local slabs, no collective ---------------------------------------------
    def score_sp(self, s: torch.Tensor, p: torch.Tensor):
        """[n, E_g]: scores of (s_i, p_i, ·) against this rank's entities."""
        rows, rel_rows = self.exchange_rows([s], p)
        return self.backend.score_emb(self.scorer, rows, rel_rows, self.ent_local, "sp_", self.l_norm)

    def score_po(self, p: torch.Tensor, o: torch.Tensor):
        if self.reciprocal_R:
            return self.score_sp(o, p + self.reciprocal_R)
        rows, rel_rows = self.exchange_rows([o], p)
        return self.backend.score_emb(self.scorer, self.ent_local, rel_rows, rows, "_po", self.l_norm)

    # a score block beyond this many bytes no longer sits in the 256 MB Infinity Cache while it is
    # written: rows are then 128-byte aligned (padded pitch) and each direction gets its own launch
    BIG_SLAB_BYTES = 96 << 20

    def score_sp_po_blocks(self, s: torch.Tensor, p: torch.Tensor, o: torch.Tensor, padded: bool = True):
        """(score_sp slab, score_po slab), each [n, E_g] (row pitch >= E_g): ONE exchange for the s
        and the o rows, then one two-sided launch on the shard (backends with score_emb_sp_po; the
        blocks are the halves of its [n, 2 E_g] output, or -- padded, the engine -- each on whole 256-byte
        lines) or, for slabs that outgrow the Infinity Cache, one launch per direction into matrices with
        a 128-byte-aligned row pitch."""
        n, m = s.numel(), self.hi - self.lo
        if self.reciprocal_R:  # two sp_ launches: (s, p) and (o, p + R); the exchanged rows are copied out of the lane's
            sp = self.score_sp(s, p)   # buffer by the first launch's consumer before the second exchange reuses it
            return sp, self.score_sp(o, p + self.reciprocal_R)
        rows, rel_rows = self.exchange_rows([s, o], p)
        s_rows, o_rows = rows[:n], rows[n:]
        big = n * m * 4 > self.BIG_SLAB_BYTES
        if hasattr(self.backend, "score_emb_sp_po") and not big:
            if padded and getattr(self.backend, "PADDED_BLOCKS", False):
                # the engine: both blocks on whole 256-byte lines (the direct-store kernel's aligned path)
                both = self.backend.score_emb_sp_po(self.scorer, s_rows, rel_rows, o_rows, self.ent_local, self.l_norm,
                                                    pad_pitch=True)
                return (both[:, 0], both[:, 1]) if both.dim() == 3 else (both[:, :m], both[:, m:])
            both = self.backend.score_emb_sp_po(self.scorer, s_rows, rel_rows, o_rows, self.ent_local, self.l_norm)
            return both[:, :m], both[:, m:]
        kw = {"pad_pitch": True} if big else {}
        return (self.backend.score_emb(self.scorer, s_rows, rel_rows, self.ent_local, "sp_", self.l_norm, **kw),
                self.backend.score_emb(self.scorer, self.ent_local, rel_rows, o_rows, "_po", self.l_norm, **kw))

    def score_sp_po(self, s: torch.Tensor, p: torch.Tensor, o: torch.Tensor):
        """[n, 2 E_g]: the two slabs of score_sp_po_blocks side by side (KgeModel.score_sp_po's layout)."""
        sp, po = self.score_sp_po_blocks(s, p, o, padded=False)  # (contiguous halves: the view below, no copy)
        if sp.data_ptr() + sp.shape[1] * 4 == po.data_ptr() and sp.stride(0) == 2 * sp.shape[1]:
            return torch.as_strided(sp, (sp.shape[0], 2 * sp.shape[1]), (sp.stride(0), 1))
        return torch.cat([sp, po], dim=1)

    # ---- training: 1vsAll cross entropy over the entities of all shards -----------------------------
    def ce_loss(self, direction: str, ids: torch.Tensor, p: torch.Tensor, labels: torch.Tensor,
                ent_master: Optional[torch.Tensor] = None, rel_master: Optional[torch.Tensor] = None,
                dropout=None, masks=None):
        """[n] cross entropy of score_sp(ids, p) ("sp": ids = subjects, labels = true objects) or
        score_po(p, ids) ("po": ids = objects, labels = true subjects) over ALL entities; sum / batch
        size = the reference's 1vsAll loss (train_1vsAll.py:64-81, loss.py:192-207).  Differentiable
        w.r.t. `ent_master` (this rank's shard of the entity parameters, [E_g, d]) and `rel_master` (the
        replicated relation parameters): float32 masters whose bf16 scoring copies are this object's
        tables (refresh them with `refresh_tables` after an optimizer step), or omitted when the tables
        themselves are the parameters."""
        ent_master = self.ent_local if ent_master is None else ent_master
        rel_master = self.rel if rel_master is None else rel_master
        if dropout is not None and max(dropout) > 0.0:
            return self._dropout_loss("ce", direction, ids, p, ent_master, rel_master, (labels,), dropout, masks)
        direction, p = self._recip(direction, p)
        return _ShardedCE.apply(self, direction, ent_master, rel_master, ids, p, labels)

    def kl_loss(self, direction: str, ids, p, lbl_rowptr, lbl_col, ent_master=None, rel_master=None, dropout=None,
                masks=None, label_smoothing: float = 0.0):
        """[n] KvsAll KL loss rows over ALL entities (see _ShardedKL); sum / batch size = the reference's loss.
        label_smoothing: KvsAll.label_smoothing (train_KvsAll.py:260-266; _kl_label_terms)."""
        ent_master = self.ent_local if ent_master is None else ent_master
        rel_master = self.rel if rel_master is None else rel_master
        eps = float(label_smoothing)
        if eps != 0.0 and self.scorer not in ("complex", "distmult"):
            raise NotImplementedError("kge_amd.sharded: label smoothing needs a scorer that is linear in the target row")
        if dropout is not None and max(dropout) > 0.0:
            return self._dropout_loss("kl", direction, ids, p, ent_master, rel_master, (lbl_rowptr, lbl_col, eps), dropout,
                                      masks)
        direction, p = self._recip(direction, p)
        return _ShardedKL.apply(self, direction, ent_master, rel_master, ids, p, lbl_rowptr, lbl_col, eps)

    def bce_loss(self, direction: str, ids, p, lbl_rowptr, lbl_col, offset: float = 0.0, ent_master=None,
                 rel_master=None, dropout=None, masks=None):
        """[n] KvsAll BCE loss rows summed over ALL entities (see _ShardedBCE)."""
        ent_master = self.ent_local if ent_master is None else ent_master
        rel_master = self.rel if rel_master is None else rel_master
        if dropout is not None and max(dropout) > 0.0:
            return self._dropout_loss("bce", direction, ids, p, ent_master, rel_master, (lbl_rowptr, lbl_col, offset),
                                      dropout, masks)
        direction, p = self._recip(direction, p)
        return _ShardedBCE.apply(self, direction, ent_master, rel_master, ids, p, lbl_rowptr, lbl_col, offset)

    def _dropout_loss(self, kind, direction, ids, p, ent_master, rel_master, extra, dropout, masks):
        """The three losses with the embedders' dropout (p_entity, p_relation) as the reference applies it in a
        training step -- masks "a" [n, d], "p" [n, d_r], "all" [E_g, d] (this rank's rows), drawn in the reference's
        order (kge_model.py:682-725) or handed in (tests).  With more than one rank the table's mask comes from a
        generator of this rank's own (`table_generator`, seeded by the caller): the default generator must see the
        same draws on every rank, and the last shard may be shorter than the others."""
        p_ent, p_rel = float(dropout[0]), float(dropout[1])
        masks = masks or {}
        gen = getattr(self, "table_generator", None) if self.world > 1 else None
        Eg = self.hi - self.lo
        shard = ent_master[:Eg]

        def drop_all():
            return _drop(shard, p_ent, masks.get("all"), gen)
        table = drop_all() if direction == "po" else None
        kdir, kp = self._recip(direction, p)
        if self.reciprocal_R and direction == "po":
            # the wrapper's own order (reciprocal_relations_model.py:84-91): all entities, relation rows p + R, o rows
            p_rows = _drop(_SumOverShards.apply(self, rel_master[kp.reshape(-1).long()]), p_rel, masks.get("p"))
            a_rows = _drop(_OwnerRows.apply(self, ent_master, ids), p_ent, masks.get("a"))
        else:
            a_rows = _drop(_OwnerRows.apply(self, ent_master, ids), p_ent, masks.get("a"))
            p_rows = _drop(_SumOverShards.apply(self, rel_master[kp.reshape(-1).long()]), p_rel, masks.get("p"))
        if table is None:
            table = drop_all()
        return _ShardedDense.apply(self, kind, kdir, table, a_rows, p_rows, extra)

    @staticmethod
    def with_slack(rows: torch.Tensor, n_max: int):
        """A shard's rows followed by 2 n_max slack rows in ONE allocation: (buffer [E_g + 2 n_max, d], view of the
        first E_g rows).  Negative-sampling training keeps its float32 master shard as that view (a Parameter over
        it shares the storage) and scores on the buffer."""
        buf = torch.zeros(rows.shape[0] + 2 * int(n_max), rows.shape[1], dtype=rows.dtype, device=rows.device)
        buf[:rows.shape[0]].copy_(rows)
        return buf, buf[:rows.shape[0]]

    def neg_scores(self, s, p, o, slot: int, neg, ent_master=None, rel_master=None):
        """(positives [n], scores [n, K] of the triples with `slot` replaced by neg[i, k]) -- see _ShardedNeg.  Needs
        `self.ent_ext` (with_slack) whose first rows are this table's `ent_local`."""
        if self.reciprocal_R:
            raise NotImplementedError("kge_amd.sharded: negative sampling over a reciprocal-relations model")
        ent_master = self.ent_local if ent_master is None else ent_master
        rel_master = self.rel if rel_master is None else rel_master
        return _ShardedNeg.apply(self, ent_master, rel_master, s, p, o, slot, neg)

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
        if (self.fused_rank and not self.reciprocal_R and K <= 2 and hasattr(self.backend, "score_rank_emb_sp_po")
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

    GRAPH = None  # None: the default below; True / False (tests)

    def __init__(self, table: ShardedEntityTable, lanes: int = 2, graph: Optional[bool] = None):
        self.table = table
        self.L = max(1, int(lanes))
        dev = table.ent_local.device
        self.cuda = dev.type == "cuda"
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(self.L)] if self.cuda and self.L > 1 else None
        self.k = 0
        self._out = []
        if graph is None:
            # default: ON for a single rank, OPT-IN (ShardedScoreLanes.GRAPH = True) for world > 1 -- the capture of a step that
            # contains a multi-GPU RCCL all-gather has run on ONE rank only in the build loop --, OFF with
            # GRAPH = False.  Either way behind a self-check in two votes (see _issue): every rank first says
            # whether its capture went through, BEFORE anything that holds a collective is replayed; only if all did
            # is the capture replayed once and compared bit for bit with the step issued call by call, and the ranks
            # vote again.  A capture that throws, differs or is voted down anywhere turns the feature off on all ranks
            # -- loudly (warnings.warn + `graph_error`) --, the step going call by call from then on.
            want = self.GRAPH
            multi = bool(table.collectives) and table.world > 1
            graph = bool(want) if want is not None else (not multi)
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
        cur = st = torch.cuda.current_stream(self.table.ent_local.device)
        if self.streams is None:  # a capture needs a stream of its own
            if self._cap_stream is None:
                self._cap_stream = torch.cuda.Stream(device=self.table.ent_local.device)
            st = self._cap_stream
        # once call by call ON THE CAPTURE STREAM: the engine's scratch is per (device, stream) and cleared once at
        # allocation -- allocated by the capture itself, that clearing fill would be a node of every replay --; the
        # communicator gets warmed up
        if st is not cur:
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                res = fn(*static)
            cur.wait_stream(st)
        else:
            res = fn(*static)
        multi = bool(tb.collectives) and tb.world > 1

        def vote(err):
            """Every rank takes the same decision: one MAX all-reduce of 'my step failed'."""
            if not multi:
                return err
            try:
                v = torch.tensor([0 if err is None else 1], device=tb.ent_local.device, dtype=torch.int32)
                dist.all_reduce(v, op=dist.ReduceOp.MAX, group=tb.group)
                if int(v.item()) != 0 and err is None:
                    err = "another rank's capture failed"
            except Exception as exc:  # pragma: no cover
                err = err or f"vote failed: {type(exc).__name__}: {exc}"
            return err

        err = None
        g = cap = None
        if tb.collectives:
            # Collectives issued so far (the warm-up call above) must be complete AND seen complete by ProcessGroupNCCL's
            # watchdog thread before the streams their end events were recorded on start capturing: HIP refuses
            # hipEventQuery on an event "last recorded in a capturing stream" also when it was recorded BEFORE the
            # capture began, and the watchdog's exception aborts the process (a fast box in round 6 lost that race:
            # profiles/r6_bench_dist1rank_watchdog_abort.txt).  The watchdog polls every 100 ms; captures are rare.
            import time
            torch.cuda.synchronize()
            time.sleep(0.35)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                cap = fn(*static)
        except Exception as exc:  # not capturable here
            err = f"{type(exc).__name__}: {exc}"
            torch.cuda.synchronize()
        # vote 1, BEFORE any replay: a capture that failed on one rank only must not leave the others replaying a
        # graph that holds a collective this rank never enters (mismatched collectives hang the job)
        err = vote(err)
        if err is None:
            try:  # self-check: one replay against the call-by-call result on the same static inputs
                g.replay()
                torch.cuda.synchronize()
                flat = lambda r: [t for t in (r if isinstance(r, (tuple, list)) else (r,)) if torch.is_tensor(t)]
                same = all(torch.equal(a, b) for a, b in zip(flat(res), flat(cap))) and len(flat(res)) == len(flat(cap))
                if not same:
                    err = "the replayed step differs from the step issued call by call"
            except Exception as exc:
                err = f"{type(exc).__name__}: {exc}"
                torch.cuda.synchronize()
            err = vote(err)  # vote 2: the comparison
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
