"""world_size-2 CPU test (gloo) of the entity-sharded path: collective choreography and
merge logic of kge_amd.sharded, with the per-shard compute provided by a FAKE backend built
on the oracle (test infrastructure; the product backend is the HIP engine and has no CPU
path).  Ranks and top-k from two shards must equal the unsharded oracle exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as ko


class _FakeTables:
    def __init__(self, scorer, ent, rel, l_norm=1.0):
        self.scorer, self.ent, self.rel, self.l_norm = scorer, ent, rel, l_norm


class OracleBackend:
    """Drop-in for kge_amd.engine in kge_amd.sharded (Tables, embed, score_emb, score_emb_sp_po,
    rank_counts, rank_counts_multi on CPU)."""

    Tables = _FakeTables

    @staticmethod
    def embed(t, ent_idx=None, rel_idx=None, ent_out=None, rel_out=None):
        """engine.embed: (ent[ent_idx], rel[rel_idx]) into the preallocated outputs."""
        if ent_idx is not None:
            ent_out.copy_(t.ent[ent_idx.long()])
        if rel_idx is not None:
            rel_out.copy_(t.rel[rel_idx.long()])
        return ent_out, rel_out

    @staticmethod
    def _scores(t, direction, a_rows, p_rows, targets):
        import torch_port as tp
        if direction == "sp":
            return tp.score_emb(t.scorer, a_rows, p_rows, targets, "sp_", t.l_norm)
        return tp.score_emb(t.scorer, targets, p_rows, a_rows, "_po", t.l_norm)

    @staticmethod
    def ce_emb_fwd(t, direction, a_rows, p_rows, label):
        """engine.ce_emb_fwd: (loss_rows -- NaN without a local label --, lse) over the shard's rows."""
        sc = OracleBackend._scores(t, direction, a_rows, p_rows, t.ent)
        lse = torch.logsumexp(sc, dim=1)
        ok = (label >= 0) & (label < t.ent.shape[0])
        true = sc.gather(1, label.clamp(0, t.ent.shape[0] - 1).view(-1, 1)).view(-1)
        return torch.where(ok, lse - true, torch.full_like(lse, float("nan"))), lse

    @staticmethod
    def ce_emb_bwd(t, direction, a_rows, p_rows, label, lse, g_rows=None, g_scalar=1.0):
        """engine.ce_emb_bwd: gradients of sum_i g_i * (lse_i - score(i, label_i)) restricted to the shard's
        columns, `lse` being the GLOBAL log-sum-exp."""
        with torch.enable_grad():  # (called from inside an autograd backward)
            a, pr, T = (x.detach().clone().requires_grad_(True) for x in (a_rows, p_rows, t.ent))
            sc = OracleBackend._scores(t, direction, a, pr, T)
            g = g_rows if g_rows is not None else torch.full_like(lse, g_scalar)
            G = torch.exp(sc.detach() - lse.view(-1, 1)) * g.view(-1, 1)
            ok = (label >= 0) & (label < T.shape[0])
            rows = torch.nonzero(ok).view(-1)
            G[rows, label[rows]] -= g[rows]
            (sc * G).sum().backward()
        return a.grad, pr.grad, T.grad

    @staticmethod
    def _label_matrix(t, rowptr, col, col_lo):
        """[n, E_g] 0/1 matrix of the labels inside the shard [col_lo, col_lo + E_g)."""
        n, Eg = rowptr.numel() - 1, t.ent.shape[0]
        y = torch.zeros(n, Eg)
        for i in range(n):
            c = col[rowptr[i]:rowptr[i + 1]] - col_lo
            c = c[(c >= 0) & (c < Eg)]
            y[i, c] = 1.0
        return y

    @staticmethod
    def kl_emb_fwd(t, direction, a_rows, p_rows, rowptr, col, col_lo, weight):
        sc = OracleBackend._scores(t, direction, a_rows, p_rows, t.ent)
        lse = torch.logsumexp(sc, dim=1)
        y = OracleBackend._label_matrix(t, rowptr, col, col_lo)
        return lse - weight * (sc * y).sum(1), lse

    @staticmethod
    def kl_emb_bwd(t, direction, a_rows, p_rows, rowptr, col, col_lo, weight, lse, g_rows=None, g_scalar=1.0,
                   label_bias=None):
        with torch.enable_grad():
            a, pr, T = (x.detach().clone().requires_grad_(True) for x in (a_rows, p_rows, t.ent))
            sc = OracleBackend._scores(t, direction, a, pr, T)
            g = g_rows if g_rows is not None else torch.full_like(lse, g_scalar)
            y = OracleBackend._label_matrix(t, rowptr, col, col_lo)
            G = torch.exp(sc.detach() - lse.view(-1, 1)) - weight.view(-1, 1) * y
            if label_bias is not None:   # label smoothing: b_i subtracted at every column (include/kge_amd.h)
                G = G - label_bias.view(-1, 1)
            G = G * g.view(-1, 1)
            (sc * G).sum().backward()
        return a.grad, pr.grad, T.grad

    @staticmethod
    def bce_emb_fwd(t, direction, a_rows, p_rows, rowptr, col, col_lo, offset=0.0):
        sc = OracleBackend._scores(t, direction, a_rows, p_rows, t.ent)
        y = OracleBackend._label_matrix(t, rowptr, col, col_lo)
        return torch.nn.functional.binary_cross_entropy_with_logits(sc + offset, y, reduction="none").sum(1)

    @staticmethod
    def bce_emb_bwd(t, direction, a_rows, p_rows, rowptr, col, col_lo, offset=0.0, g_rows=None, g_scalar=1.0):
        with torch.enable_grad():
            a, pr, T = (x.detach().clone().requires_grad_(True) for x in (a_rows, p_rows, t.ent))
            sc = OracleBackend._scores(t, direction, a, pr, T)
            g = g_rows if g_rows is not None else torch.full((sc.shape[0],), g_scalar)
            y = OracleBackend._label_matrix(t, rowptr, col, col_lo)
            G = (torch.sigmoid(sc.detach() + offset) - y) * g.view(-1, 1)
            (sc * G).sum().backward()
        return a.grad, pr.grad, T.grad

    # negative sampling on the table + slack rows (index-level, as kge_score_spo / kge_score_neg and their twins)
    @staticmethod
    def _spo(t, ent, rel, s, p, o):
        import torch_port as tp
        return tp.score_emb(t.scorer, ent[s], rel[p], ent[o], "spo", t.l_norm).view(-1)

    @staticmethod
    def score_spo(t, s, p, o):
        return OracleBackend._spo(t, t.ent, t.rel, s.long(), p.long(), o.long()).detach()

    @staticmethod
    def _neg(t, ent, rel, s, p, o, slot, neg):
        n, K = neg.shape
        rep = lambda x: x.long().view(-1, 1).expand(n, K).reshape(-1)
        ss, oo = (neg.reshape(-1), rep(o)) if slot == 0 else (rep(s), neg.reshape(-1))
        return OracleBackend._spo(t, ent, rel, ss, rep(p), oo).view(n, K)

    @staticmethod
    def score_neg(t, s, p, o, slot, neg):
        return OracleBackend._neg(t, t.ent, t.rel, s, p, o, slot, neg).detach()

    @staticmethod
    def score_neg_bwd_accum(t, s, p, o, slot, neg, gout, scores, grad_ent, grad_rel):
        with torch.enable_grad():
            ent, rel = (x.detach().clone().requires_grad_(True) for x in (t.ent, t.rel))
            (OracleBackend._neg(t, ent, rel, s, p, o, slot, neg) * gout).sum().backward()
        grad_ent += ent.grad
        grad_rel += rel.grad
        return True

    @staticmethod
    def score_spo_bwd_accum(t, s, p, o, gout, scores, grad_ent, grad_rel):
        with torch.enable_grad():
            ent, rel = (x.detach().clone().requires_grad_(True) for x in (t.ent, t.rel))
            (OracleBackend._spo(t, ent, rel, s.long(), p.long(), o.long()) * gout).sum().backward()
        grad_ent += ent.grad
        grad_rel += rel.grad

    @staticmethod
    def score_emb_sp_po(scorer, s_emb, p_emb, o_emb, targets, l_norm=1.0):
        return torch.cat([OracleBackend.score_emb(scorer, s_emb, p_emb, targets, "sp_", l_norm),
                          OracleBackend.score_emb(scorer, targets, p_emb, o_emb, "_po", l_norm)], 1)

    @staticmethod
    def score_emb(scorer, s_emb, p_emb, o_emb, combine, l_norm=1.0):
        s_emb, p_emb, o_emb = (x.detach().numpy() for x in (s_emb, p_emb, o_emb))
        n = p_emb.shape[0]
        if combine == "sp_":
            t = ko.Tables(scorer, np.concatenate([s_emb, o_emb]), p_emb, l_norm)
            out = ko.score_sp(t, np.arange(n), np.arange(n), n + np.arange(o_emb.shape[0]))
        elif combine == "_po":
            t = ko.Tables(scorer, np.concatenate([o_emb, s_emb]), p_emb, l_norm)
            out = ko.score_po(t, np.arange(n), np.arange(n), n + np.arange(s_emb.shape[0]))
        else:
            raise ValueError(combine)
        return torch.from_numpy(out)

    @staticmethod
    def rank_counts(scores, true, rp=None, col=None, col_offset=0, true_col=None, atol=1e-5, rtol=1e-4):
        conv = lambda x: None if x is None else x.numpy()
        r, t = ko.rank_counts(scores.numpy(), true.numpy(), conv(rp), conv(col), col_offset,
                              conv(true_col), atol, rtol)
        return torch.from_numpy(r), torch.from_numpy(t)


    @staticmethod
    def rank_counts_multi(scores, true, filters, col_offset, true_col, atol, rtol, rank, ties):
        """kge_rank_counts_multi on the oracle: one rank_counts per ranking (rank / ties [K + 1, n])."""
        sc, tr, tc = scores.numpy(), true.numpy(), true_col.numpy()
        r, t = ko.rank_counts(sc, tr, atol=atol, rtol=rtol)
        rank[0] += torch.from_numpy(r)
        ties[0] += torch.from_numpy(t)
        for k, (beg, end, vals) in enumerate(filters):
            beg, end, vals = beg.numpy(), end.numpy(), vals.numpy()
            rp = np.concatenate([[0], np.cumsum(end - beg)]).astype(np.int64)
            col = np.concatenate([vals[b:e] for b, e in zip(beg, end)] + [np.zeros(0, np.int64)]).astype(np.int64)
            r, t = ko.rank_counts(sc, tr, rp, col, col_offset, tc, atol, rtol)
            rank[k + 1] += torch.from_numpy(r)
            ties[k + 1] += torch.from_numpy(t)


class FusedOracleBackend(OracleBackend):
    """+ engine.score_rank_emb_sp_po (counts of this shard's columns without a score slab leaving the call),
    restated on the oracle: the choreography of ShardedEntityTable._rank_batch_fused runs on CPU."""
    calls = 0

    @staticmethod
    def score_rank_emb_sp_po(scorer, s_emb, p_emb, o_emb, s_ids, o_ids, targets, col_begin, true_sp, true_po,
                             filters_sp, filters_po, atol, rtol, rank_sp, ties_sp, rank_po, ties_po, l_norm=1.0):
        FusedOracleBackend.calls += 1
        m = targets.shape[0]
        both = OracleBackend.score_emb_sp_po(scorer, s_emb, p_emb, o_emb, targets, l_norm)
        OracleBackend.rank_counts_multi(both[:, :m], true_sp, filters_sp, col_begin, o_ids.long(), atol, rtol,
                                        rank_sp, ties_sp)
        OracleBackend.rank_counts_multi(both[:, m:], true_po, filters_po, col_begin, s_ids.long(), atol, rtol,
                                        rank_po, ties_po)
        return True



def _quiet_teardown():
    """Every rank reaches this point before any rank closes its sockets: a rank that tears its gloo context down while the
    other is still inside its last collective aborts the straggler ("terminate called without an active exception": one
    run in six on this box before the barrier was here)."""
    try:
        if dist.is_initialized():
            import datetime
            dist.monitored_barrier(timeout=datetime.timedelta(seconds=30))  # (bounded: the other rank may have died)
    except Exception:
        pass
    if dist.is_initialized():
        dist.destroy_process_group()

def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, model, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kge_amd.eval import FilterIndex
        from kge_amd.sharded import ShardedEntityTable
        from kge_amd.synthetic import make_splits

        E, R, d = 53, 4, 16   # odd E: ragged shards
        rng = np.random.default_rng(0)
        ent = rng.standard_normal((E, d)).astype(np.float32)
        rel = (rng.uniform(-3, 3, (R, d // 2)) if model == "rotate" else rng.standard_normal((R, d))).astype(np.float32)
        splits = make_splits(E, R, 300, 40, 40, seed=2)
        lo, hi = ShardedEntityTable.partition(E, world, rank)
        sh = ShardedEntityTable(model, torch.from_numpy(ent[lo:hi]), torch.from_numpy(rel), E,
                                backend=OracleBackend)
        batch = splits["valid"][:24]
        fi = FilterIndex([splits["train"], splits["valid"]], E, R)
        labels = tuple(torch.from_numpy(x) for x in fi.labels(batch))
        tb = torch.from_numpy(batch.astype(np.int64))
        out = {}
        for key, lab in (("raw", None), ("filt", labels)):
            s_rank, s_ties, o_rank, o_ties = sh.rank_batch(tb, lab)
            out[key] = (s_rank.numpy(), s_ties.numpy(), o_rank.numpy(), o_ties.numpy())
        # raw + filtered from one call (device-resident index ranges instead of a per-batch CSR)
        sb, se, pb, pe = (torch.from_numpy(x) for x in fi.ranges(batch))
        cm = sh.rank_batch_multi(tb, [(sb, se, torch.from_numpy(fi.sp_values))],
                                 [(pb, pe, torch.from_numpy(fi.po_values))])
        for k, key in enumerate(("raw", "filt")):
            s_rank, s_ties, o_rank, o_ties = out[key]
            assert np.array_equal(cm[0, 0, k].numpy(), o_rank) and np.array_equal(cm[0, 1, k].numpy(), o_ties), key
            assert np.array_equal(cm[1, 0, k].numpy(), s_rank) and np.array_equal(cm[1, 1, k].numpy(), s_ties), key
        # the same counts with the backend's score + rank entry: true scores from the exchanged rows on every rank
        # (no all-reduce of their own), shard-local counts, ONE counter all-reduce
        sh2 = ShardedEntityTable(model, torch.from_numpy(ent[lo:hi]), torch.from_numpy(rel), E,
                                 backend=FusedOracleBackend)
        cm2 = sh2.rank_batch_multi(tb, [(sb, se, torch.from_numpy(fi.sp_values))],
                                   [(pb, pe, torch.from_numpy(fi.po_values))])
        assert FusedOracleBackend.calls == 1 and sh2.fused_rank
        assert torch.equal(cm2, cm)
        # the whole evaluation loop over the sharded table (EntityRankingEvaluator._run_sharded): replicated filter
        # index, shard-local counts, one counter all-reduce per batch.  Its two device-side helpers (filter lookup,
        # tie policy + histogram) restated in numpy for this CPU run.
        import kge_amd.eval as kev
        TIES = {"rounded_mean_rank": lambda r, t: r + t // 2, "best_rank": lambda r, t: r, "worst_rank": lambda r, t: r + t - 1}

        def lookup_multi(queries):
            for keys, starts, a, b, mult, beg, end in queries:
                k, st_ = keys.numpy(), starts.numpy()
                q = a.long().numpy() * mult + b.long().numpy()
                pos = np.minimum(np.searchsorted(k, q), max(len(k) - 1, 0))
                hit = (k[pos] == q) if len(k) else np.zeros(len(q), bool)
                beg.copy_(torch.from_numpy(np.where(hit, st_[pos], 0)))
                end.copy_(torch.from_numpy(np.where(hit, st_[pos + 1] if len(k) else 0, 0)))

        def hist_update(rank, ties, policy, hist, ranks_out=None):
            r = TIES[policy](rank, ties)
            for m in range(r.shape[0]):
                hist[m].index_add_(0, r[m], torch.ones(r.shape[1]))
            if ranks_out is not None:
                ranks_out.copy_(r)

        orig = kev.engine.filter_lookup_multi, kev.engine.rank_hist
        kev.engine.filter_lookup_multi, kev.engine.rank_hist = lookup_multi, hist_update
        try:
            ev = kev.EntityRankingEvaluator(sh2, splits, E, R, eval_split="valid", batch_size=17)
            ev_metrics, ev_ranks = ev.run(return_ranks=True)
        finally:
            kev.engine.filter_lookup_multi, kev.engine.rank_hist = orig
        rows = sh.gather_entity_rows(tb[:, 0])
        assert np.array_equal(rows.numpy(), ent[batch[:, 0]])
        # the call sequence of bench.py --gpus N: one exchange for the s and o rows (strided int32
        # views of the batch, as the trainers pass them), relation rows from the replicated table,
        # one two-sided scoring call on the shard -> this rank's [n, 2 E_g] slab
        b32 = torch.from_numpy(batch.astype(np.int32))
        rows2, rel_rows = sh.exchange_rows([b32[:, 0], b32[:, 2]], b32[:, 1])
        assert np.array_equal(rows2.numpy(), np.concatenate([ent[batch[:, 0]], ent[batch[:, 2]]]))
        assert np.array_equal(rel_rows.numpy(), rel[batch[:, 1]])
        both = sh.score_sp_po(b32[:, 0], b32[:, 1], b32[:, 2])
        full = ko.Tables(model, ent, rel, 1.0)
        want = np.concatenate([ko.score_sp(full, batch[:, 0], batch[:, 1], np.arange(lo, hi)),
                               ko.score_po(full, batch[:, 1], batch[:, 2], np.arange(lo, hi))], 1)
        assert np.array_equal(both.numpy(), want)
        # several batches in flight (ShardedScoreLanes; on CPU: the lanes' separate exchange buffers, no streams):
        # three different batches through two lanes = the three direct calls, read after join()
        from kge_amd.sharded import ShardedScoreLanes
        lanes = ShardedScoreLanes(sh, 2)
        parts = [torch.from_numpy(splits["valid"][k * 12:(k + 1) * 12].astype(np.int64)) for k in range(3)]
        lanes.fork()
        res = [lanes.score_sp_po_blocks(x[:, 0], x[:, 1], x[:, 2]) for x in parts]
        lanes.join()
        for x, (a_sp, a_po) in zip(parts, res):
            w_sp, w_po = sh.score_sp_po_blocks(x[:, 0], x[:, 1], x[:, 2])
            assert torch.equal(a_sp, w_sp) and torch.equal(a_po, w_po)
        slab = sh.score_sp(tb[:, 0], tb[:, 1])
        tv, ti = sh.topk(slab, 5)
        if rank == 0:
            q.put((out, tv.numpy(), ti.numpy(), ent, rel, splits, batch, ev_metrics, ev_ranks))
    finally:
        _quiet_teardown()


@pytest.mark.parametrize("model", ["complex", "transe", "rotate"])
def test_two_shards_equal_unsharded(model):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, model, q)) for r in range(world)]
    for p in procs:
        p.start()
    import time
    got, t0 = None, time.time()
    while got is None and time.time() - t0 < 240:  # a crashed worker must fail the test, not hang it
        if not q.empty():
            got = q.get()
        elif any(p.exitcode not in (None, 0) for p in procs):
            break
        else:
            time.sleep(0.05)
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    assert got is not None
    out, tv, ti, ent, rel, splits, batch, ev_metrics, ev_ranks = got
    E = ent.shape[0]
    t = ko.Tables(model, ent, rel, 1.0)
    s, p_, o = batch[:, 0], batch[:, 1], batch[:, 2]
    sp, po = ko.score_sp(t, s, p_), ko.score_po(t, p_, o)
    o_true, s_true = sp[np.arange(len(s)), o], po[np.arange(len(s)), s]
    ix = [(ko.build_index(splits[k], (0, 1), 2), ko.build_index(splits[k], (1, 2), 0)) for k in ("train", "valid")]
    for key in ("raw", "filt"):
        if key == "raw":
            kw_o, kw_s = {}, {}
        else:
            rp, col = ko.labels_csr(batch[:, [0, 1]], [i[0] for i in ix])
            kw_o = dict(lbl_rowptr=rp, lbl_col=col, true_col=o)
            rp, col = ko.labels_csr(batch[:, [1, 2]], [i[1] for i in ix])
            kw_s = dict(lbl_rowptr=rp, lbl_col=col, true_col=s)
        o_rank, o_ties = ko.rank_counts(sp, o_true, **kw_o)
        s_rank, s_ties = ko.rank_counts(po, s_true, **kw_s)
        g = out[key]
        assert np.array_equal(g[0], s_rank) and np.array_equal(g[1], s_ties), key
        assert np.array_equal(g[2], o_rank) and np.array_equal(g[3], o_ties), key
    # the sharded evaluation loop: per-example ranks of all three rankings = the oracle's restatement of
    # EntityRankingJob._evaluate over the whole valid split, metrics from the summed histograms
    valid = splits["valid"].astype(np.int64)
    tsp, tpo = ko.build_index(splits["test"], (0, 1), 2), ko.build_index(splits["test"], (1, 2), 0)
    isp, ipo = [i[0] for i in ix], [i[1] for i in ix]
    for key, a, b in (("_raw", None, None), ("_filt", isp, ipo), ("_filt_test", isp + [tsp], ipo + [tpo])):
        s_r, o_r = ko.evaluate_ranks(t, valid, a, b)
        assert np.array_equal(ev_ranks["o" + key], o_r), ("sharded evaluator", "o" + key)
        assert np.array_equal(ev_ranks["s" + key], s_r), ("sharded evaluator", "s" + key)
    both_r = np.concatenate([ev_ranks["s_filt"], ev_ranks["o_filt"]]).astype(np.float64) + 1.0
    assert abs(ev_metrics["mean_reciprocal_rank_filtered"] - float(np.mean(1.0 / both_r))) <= 1e-6
    order = np.argsort(-sp, axis=1, kind="stable")[:, :5]
    assert np.array_equal(np.take_along_axis(sp, order, 1), tv)
    assert np.array_equal(np.sort(ti, 1), np.sort(order, 1)) or np.allclose(np.take_along_axis(sp, ti, 1), tv)


def _train_worker(rank, world, port, model, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kge_amd.sharded import ShardedEntityTable
        E, R, d, n = 47, 5, 16, 21   # odd E: ragged shards
        g = torch.Generator().manual_seed(3)
        ent = torch.randn(E, d, generator=g)
        rel = torch.randn(R, d, generator=g)
        s, p, o = (torch.randint(hi, (n,), generator=g) for hi in (E, R, E))
        w = torch.rand(2 * n, generator=g) + 0.5  # per-row weights: a non-trivial upstream gradient
        lo, hi = ShardedEntityTable.partition(E, world, rank)
        ent_master = ent[lo:hi].clone().requires_grad_(True)   # this rank's shard of the parameters
        rel_master = rel.clone().requires_grad_(True)          # replicated
        sh = ShardedEntityTable(model, ent_master.detach().clone(), rel_master.detach().clone(), E,
                                backend=OracleBackend)
        # one 1vsAll step (train_1vsAll.py:64-81): sp_ rows labelled with o, _po rows labelled with s
        loss_sp = sh.ce_loss("sp", s, p, o, ent_master, rel_master)
        loss_po = sh.ce_loss("po", o, p, s, ent_master, rel_master)
        (torch.cat([loss_sp, loss_po]) * w).sum().backward()
        q.put((rank, lo, hi, loss_sp.detach().numpy(), loss_po.detach().numpy(), ent_master.grad.numpy(),
               rel_master.grad.numpy(), ent.numpy(), rel.numpy(), s.numpy(), p.numpy(), o.numpy(), w.numpy()))
    finally:
        _quiet_teardown()


@pytest.mark.parametrize("model", ["complex", "distmult"])
def test_sharded_1vsAll_loss_and_gradients_equal_unsharded(model):
    """Entity-sharded 1vsAll training step on two ranks (ShardedEntityTable.ce_loss: per-shard fused
    score + loss, log-sum-exps merged across the shards, query-row gradients summed with one
    all-reduce, target-row gradients local) against the unsharded computation: torch cross entropy of
    the reference's op sequence over all entities and autograd of it -- per-row losses, the entity
    gradient (the two shards' rows side by side) and the relation gradient, which every rank must hold
    identically."""
    import torch.nn.functional as F
    import torch_port as tp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, model, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    outs = []
    import time
    t0 = time.time()
    while len(outs) < world and time.time() - t0 < 120:  # a crashed worker must fail the test, not hang it
        if not q.empty():
            outs.append(q.get())
        elif any(pr.exitcode not in (None, 0) for pr in procs):
            break
        else:
            time.sleep(0.05)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert len(outs) == world
    outs.sort(key=lambda x: x[0])
    _, _, _, lsp, lpo, _, _, ent, rel, s, p, o, w = outs[0]
    ent_t, rel_t = torch.from_numpy(ent).requires_grad_(True), torch.from_numpy(rel).requires_grad_(True)
    s, p, o, w = (torch.from_numpy(x) for x in (s, p, o, w))
    ref_sp = F.cross_entropy(tp.score_sp(model, ent_t, rel_t, s, p), o, reduction="none")
    ref_po = F.cross_entropy(tp.score_po(model, ent_t, rel_t, p, o), s, reduction="none")
    (torch.cat([ref_sp, ref_po]) * w).sum().backward()
    for out in outs:  # losses are global on every rank
        np.testing.assert_allclose(out[3], ref_sp.detach().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(out[4], ref_po.detach().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(out[6], rel_t.grad.numpy(), rtol=1e-4, atol=1e-5)  # relation grads: same everywhere
    ge = np.concatenate([out[5] for out in outs])
    assert [out[1] for out in outs] == [0, outs[0][2]] and outs[-1][2] == ent.shape[0]
    np.testing.assert_allclose(ge, ent_t.grad.numpy(), rtol=1e-4, atol=1e-5)


def _job_worker(rank, world, port, model, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kge_amd.sharded_train import ShardedTrainingJob1vsAll
        E, R, d, n = 47, 5, 16, 21
        g = torch.Generator().manual_seed(9)
        batches = [torch.stack([torch.randint(hi, (n,), generator=g) for hi in (E, R, E)], 1) for _ in range(3)]
        job = ShardedTrainingJob1vsAll(model, E, R, d, seed=4, lr=0.3, optimizer="Adagrad", score_dtype=torch.float32,
                                       backend=OracleBackend)
        losses = [float(job.step(batches[0])), float(job.step(batches[1]))]
        ck = job.checkpoint()                       # after two steps: ONE [E, d] parameter + gathered optimizer state
        losses.append(float(job.step(batches[2])))
        sd = job.state_dict()
        # a job resumed from the checkpoint on this world size takes the same third step
        job2 = ShardedTrainingJob1vsAll(model, E, R, d, seed=99, lr=0.3, optimizer="Adagrad",
                                        score_dtype=torch.float32, backend=OracleBackend)
        job2.load_checkpoint(ck)
        l3 = float(job2.step(batches[2]))
        sd2 = job2.state_dict()
        assert ck["type"] == "train" and "valid_trace" in ck and isinstance(ck["model"], tuple)   # TrainingJob.save_to's keys
        q.put((rank, losses, {k: v.numpy() for k, v in sd.items()}, {k: v.numpy() for k, v in ck["model"][0].items()},
               {pid: {kk: (vv.numpy() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                for pid, v in ck["optimizer_state_dict"]["state"].items()}, l3, {k: v.numpy() for k, v in sd2.items()},
               ck["optimizer_state_dict"]["param_groups"]))
    finally:
        _quiet_teardown()


@pytest.mark.parametrize("model", ["complex", "distmult"])
def test_sharded_training_job_equals_the_unsharded_run(model):
    """ShardedTrainingJob1vsAll on two ranks, three optimizer steps (Adagrad on each rank's own rows, replicated
    relation table stepping in lock-step without an all-reduce): avg_loss per batch and the gathered parameters equal
    the unsharded run -- the reference's 1vsAll step (cross entropy `sum` / batch size of score_sp and score_po over
    all entities, train_1vsAll.py:64-81) with torch.optim.Adagrad on the full tables.  The checkpoint after two steps
    holds ONE [E, d] entity parameter under the reference's name, and a job resumed from it -- here on the same two
    ranks, below on ONE rank -- takes the same third step."""
    import torch.nn.functional as F
    import torch_port as tp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_job_worker, args=(r, world, port, model, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    outs = []
    import time
    t0 = time.time()
    while len(outs) < world and time.time() - t0 < 180:
        if not q.empty():
            outs.append(q.get())
        elif any(pr.exitcode not in (None, 0) for pr in procs):
            break
        else:
            time.sleep(0.05)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert len(outs) == world
    outs.sort(key=lambda x: x[0])
    # the unsharded run
    E, R, d, n = 47, 5, 16, 21
    g = torch.Generator().manual_seed(4)
    ent = torch.empty(E, d).normal_(0.0, 0.1, generator=g).requires_grad_(True)
    rel = torch.empty(R, d).normal_(0.0, 0.1, generator=g).requires_grad_(True)
    g = torch.Generator().manual_seed(9)
    batches = [torch.stack([torch.randint(hi, (n,), generator=g) for hi in (E, R, E)], 1) for _ in range(3)]
    opt = torch.optim.Adagrad([ent, rel], lr=0.3)
    ref_losses, ref_after2 = [], None
    for k, b in enumerate(batches):
        s, p, o = b[:, 0], b[:, 1], b[:, 2]
        opt.zero_grad()
        l_sp = F.cross_entropy(tp.score_sp(model, ent, rel, s, p), o, reduction="sum") / n
        l_po = F.cross_entropy(tp.score_po(model, ent, rel, p, o), s, reduction="sum") / n
        (l_sp + l_po).backward()
        opt.step()
        ref_losses.append(float(l_sp + l_po))
        if k == 1:
            ref_after2 = (ent.detach().clone().numpy(), rel.detach().clone().numpy(),
                          opt.state[ent]["sum"].clone().numpy())
    from kge_amd.sharded_train import ENT_KEY, REL_KEY, ShardedTrainingJob1vsAll
    for rank, losses, sd, ck_sd, ck_opt, l3, sd2, ck_groups in outs:
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-5, atol=1e-6)
        assert sd[ENT_KEY].shape == (E, d) and ck_sd[ENT_KEY].shape == (E, d) and ck_opt[0]["sum"].shape == (E, d)
        np.testing.assert_allclose(sd[ENT_KEY], ent.detach().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(sd[REL_KEY], rel.detach().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(ck_sd[ENT_KEY], ref_after2[0], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(ck_opt[0]["sum"], ref_after2[2], rtol=1e-4, atol=1e-7)
        assert abs(l3 - losses[2]) <= 1e-6 * max(1.0, abs(l3))
        np.testing.assert_allclose(sd2[ENT_KEY], sd[ENT_KEY], rtol=1e-6, atol=1e-7)
    assert np.array_equal(outs[0][2][ENT_KEY], outs[1][2][ENT_KEY])  # the gathered parameter is the same everywhere
    # the two-rank checkpoint resumed on ONE rank (no process group): the same third step
    rank, losses, sd, ck_sd, ck_opt, l3, sd2, ck_groups = outs[0]
    opt_sd = {"state": {pid: {kk: (torch.from_numpy(vv) if isinstance(vv, np.ndarray) else vv) for kk, vv in v.items()}
                        for pid, v in ck_opt.items()}, "param_groups": ck_groups}
    # the optimizer state is torch's own layout: a reference-style UNSHARDED torch.optim.Adagrad over [entities,
    # relations] (kge/util/optimizer.py:15-20) loads it as is and holds the reference run's accumulator
    e2, r2 = torch.zeros(E, d, requires_grad=True), torch.zeros(R, d, requires_grad=True)
    ref_opt = torch.optim.Adagrad([e2, r2], lr=0.3)
    ref_opt.load_state_dict(opt_sd)
    np.testing.assert_allclose(ref_opt.state[e2]["sum"].numpy(), ref_after2[2], rtol=1e-4, atol=1e-7)
    one = ShardedTrainingJob1vsAll(model, E, R, d, seed=1, lr=0.3, optimizer="Adagrad", score_dtype=torch.float32,
                                   backend=OracleBackend)
    one.load_checkpoint({"type": "train", "epoch": 2, "valid_trace": [],
                         "model": ({k: torch.from_numpy(v) for k, v in ck_sd.items()}, {}),
                         "optimizer_state_dict": opt_sd})
    l3_one = float(one.step(batches[2]))
    assert abs(l3_one - losses[2]) <= 1e-5 * max(1.0, abs(l3_one))
    np.testing.assert_allclose(one.state_dict()[ENT_KEY].numpy(), sd[ENT_KEY], rtol=1e-4, atol=1e-6)


def _labels(g, n, E, kmax=5):
    """A label CSR of global entity ids, some rows empty (TrainingJobKvsAll's batch["label_coords"] as CSR)."""
    cnt = torch.randint(0, kmax + 1, (n,), generator=g)
    cnt[0] = 0
    rowptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(cnt, 0)])
    col = torch.cat([torch.randperm(E, generator=g)[:int(c)] for c in cnt] + [torch.zeros(0, dtype=torch.int64)])
    return rowptr, col


def _kvs_ns_worker(rank, world, port, model, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kge_amd.sharded_train import ENT_KEY, ShardedTrainingJobKvsAll, ShardedTrainingJobNegativeSampling
        E, R, d, n, K = 47, 5, 16, 13, 7
        out = {}
        for loss in ("kl", "bce"):
            g = torch.Generator().manual_seed(21)
            job = ShardedTrainingJobKvsAll(model, E, R, d, seed=4, lr=0.5, optimizer="SGD", score_dtype=torch.float32,
                                           backend=OracleBackend, loss=loss, loss_arg=0.25 if loss == "bce" else 0.0)
            losses = []
            for _ in range(3):
                qs = []
                for direction in ("sp", "po"):
                    ids, p = torch.randint(E, (n,), generator=g), torch.randint(R, (n,), generator=g)
                    qs.append((direction, ids, p) + _labels(g, n, E))
                losses.append(float(job.step(qs)))
            ck = job.checkpoint()
            job2 = ShardedTrainingJobKvsAll(model, E, R, d, seed=5, lr=0.5, optimizer="SGD", score_dtype=torch.float32,
                                            backend=OracleBackend, loss=loss, loss_arg=0.25 if loss == "bce" else 0.0)
            job2.load_checkpoint(ck)
            assert np.array_equal(job2.state_dict()[ENT_KEY].numpy(), job.state_dict()[ENT_KEY].numpy())
            out[loss] = (losses, job.state_dict()[ENT_KEY].numpy())
        for loss in ("kl", "bce"):
            g = torch.Generator().manual_seed(22)
            job = ShardedTrainingJobNegativeSampling(model, E, R, d, seed=4, lr=0.5, optimizer="SGD", n_max=n,
                                                     backend=OracleBackend, loss=loss)
            losses = []
            for _ in range(3):
                tri = torch.stack([torch.randint(hi, (n,), generator=g) for hi in (E, R, E)], 1)
                ns, no = torch.randint(E, (n, K), generator=g), torch.randint(E, (n, K - 2), generator=g)
                losses.append(float(job.step(tri, ns, no)))
            out["ns_" + loss] = (losses, job.state_dict()[ENT_KEY].numpy())
        q.put((rank, out))
    finally:
        _quiet_teardown()


@pytest.mark.parametrize("model", ["complex", "transe"])
def test_sharded_kvsall_and_negative_sampling_jobs_equal_the_unsharded_runs(model):
    """ShardedTrainingJobKvsAll (train.loss kl / bce) and ShardedTrainingJobNegativeSampling (kl / bce) on two ranks,
    three SGD steps each (a step linear in the gradient: Adagrad's first steps divide by
    |g| + 1e-10 and turn float32 summation noise on near-zero gradients into visible parameter differences; the Adagrad
    path and the checkpoints are the 1vsAll test's), against the unsharded reference steps: TrainingJobKvsAll._process_subbatch
    (train_KvsAll.py:274-294: score_sp / score_po against all entities, KLDivWithSoftmaxKgeLoss on the normalised
    multi-hot labels / BCEWithLogitsKgeLoss, sum / number of queries) and TrainingJobNegativeSampling._process_subbatch
    (train_negative_sampling.py:120-163: per slot the [n, 1 + K] block, loss with label 0, sum / n) with
    torch.optim.SGD on the full tables.  Losses per step and the gathered entity table after three steps."""
    import time
    import torch.nn.functional as F
    import torch_port as tp
    if model == "transe":  # (the fused KvsAll losses are ComplEx / DistMult kernels; the fake backend scores any model)
        pass
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_kvs_ns_worker, args=(r, world, port, model, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    outs, t0 = [], time.time()
    while len(outs) < world and time.time() - t0 < 240:
        if not q.empty():
            outs.append(q.get())
        elif any(pr.exitcode not in (None, 0) for pr in procs):
            break
        else:
            time.sleep(0.05)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert len(outs) == world
    E, R, d, n, K = 47, 5, 16, 13, 7

    def fresh():
        g = torch.Generator().manual_seed(4)
        ent = torch.empty(E, d).normal_(0.0, 0.1, generator=g).requires_grad_(True)
        rel = torch.empty(R, d).normal_(0.0, 0.1, generator=g).requires_grad_(True)
        return ent, rel, torch.optim.SGD([ent, rel], lr=0.5)

    ref = {}
    for loss in ("kl", "bce"):
        ent, rel, opt = fresh()
        g = torch.Generator().manual_seed(21)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            total = 0.0
            groups = []
            for direction in ("sp", "po"):
                ids, p = torch.randint(E, (n,), generator=g), torch.randint(R, (n,), generator=g)
                groups.append((direction, ids, p) + _labels(g, n, E))
            for direction, ids, p, rowptr, col in groups:
                sc = tp.score_sp(model, ent, rel, ids, p) if direction == "sp" else tp.score_po(model, ent, rel, p, ids)
                y = torch.zeros(n, E)
                for i in range(n):
                    y[i, col[rowptr[i]:rowptr[i + 1]]] = 1.0
                if loss == "kl":
                    yn = F.normalize(y, p=1, dim=1)   # loss.py:208-213
                    l = F.kl_div(F.log_softmax(sc, dim=1), yn, reduction="sum") / (2 * n)
                else:
                    l = F.binary_cross_entropy_with_logits(sc + 0.25, y, reduction="sum") / (2 * n)
                l.backward()
                total += float(l)
            opt.step()
            losses.append(total)
        ref[loss] = (losses, ent.detach().numpy().copy())
    for loss in ("kl", "bce"):
        ent, rel, opt = fresh()
        g = torch.Generator().manual_seed(22)
        losses = []
        for _ in range(3):
            tri = torch.stack([torch.randint(hi, (n,), generator=g) for hi in (E, R, E)], 1)
            ns, no = torch.randint(E, (n, K), generator=g), torch.randint(E, (n, K - 2), generator=g)
            s, p, o = tri[:, 0], tri[:, 1], tri[:, 2]
            opt.zero_grad()
            total = 0.0
            for slot, neg in ((0, ns), (2, no)):
                kk = neg.shape[1]
                rep = lambda x: x.view(-1, 1).expand(n, kk).reshape(-1)
                pos = tp.score_emb(model, ent[s], rel[p], ent[o], "spo").view(-1)
                ss, oo = (neg.reshape(-1), rep(o)) if slot == 0 else (rep(s), neg.reshape(-1))
                neg_sc = tp.score_emb(model, ent[ss], rel[rep(p)], ent[oo], "spo").view(n, kk)
                block = torch.cat([pos.view(-1, 1), neg_sc], 1)
                if loss == "kl":
                    l = F.cross_entropy(block, torch.zeros(n, dtype=torch.long), reduction="sum") / n
                else:
                    lab = torch.zeros_like(block)
                    lab[:, 0] = 1.0
                    l = F.binary_cross_entropy_with_logits(block, lab, reduction="sum") / n
                l.backward()
                total += float(l)
            opt.step()
            losses.append(total)
        ref["ns_" + loss] = (losses, ent.detach().numpy().copy())
    for rank, out in outs:
        for key, (losses, ent_after) in out.items():
            np.testing.assert_allclose(losses, ref[key][0], rtol=2e-5, atol=1e-6, err_msg=key)
            np.testing.assert_allclose(ent_after, ref[key][1], rtol=2e-4, atol=2e-6, err_msg=key)


def test_a_split_that_leaves_a_rank_without_rows_is_refused_on_every_rank():
    """E = 9 over 4 ranks = 3 rows each, rank 3 owns [9, 9).  The kernels refuse an empty table, and one rank
    raising alone leaves the others in the next all-gather; the split depends on (E, world) only, so the constructor
    refuses it everywhere.  (The `isfinite` guard on the kl labels in _ShardedKL stays: a shard's lse may also be -inf
    because all its scores are.)"""
    from kge_amd.sharded import ShardedEntityTable
    from kge_amd.sharded_train import ShardedTrainingJob1vsAll
    with pytest.raises(ValueError, match="without rows"):
        ShardedEntityTable.check_partition(9, 4)
    with pytest.raises(ValueError, match="without rows"):
        ShardedEntityTable.check_partition(1, 2)
    for E, world in ((9, 3), (10, 4), (9, 9), (14541, 8), (4818298, 8), (7, 1)):
        ShardedEntityTable.check_partition(E, world)
        parts = [ShardedEntityTable.partition(E, world, r) for r in range(world)]
        assert all(hi > lo for lo, hi in parts) and parts[0][0] == 0 and parts[-1][1] == E
        assert all(parts[r][1] == parts[r + 1][0] for r in range(world - 1))
    with pytest.raises(ValueError, match="without rows"):  # (no process group: one rank, zero entities)
        ShardedTrainingJob1vsAll("distmult", 0, 3, 8, backend=OracleBackend)


# ---- embedder dropout over the sharded table (round 6) ---------------------------------------------------------------
def _dropout_worker(rank, world, port, model, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kge_amd.sharded import ShardedEntityTable
        E, R, d, n = 47, 5, 16, 21
        g = torch.Generator().manual_seed(5)
        ent = torch.randn(E, d, generator=g)
        rel = torch.randn(R, d, generator=g)
        s, p, o = (torch.randint(hi, (n,), generator=g) for hi in (E, R, E))
        w = torch.rand(n, generator=g) + 0.5
        rowptr, col = _labels(g, n, E)
        pe, pr = 0.3, 0.2
        m_all, m_a, m_p = ((torch.rand(shape, generator=g) >= prob).float()
                           for shape, prob in (((E, d), pe), ((n, d), pe), ((n, d), pr)))
        lo, hi = ShardedEntityTable.partition(E, world, rank)
        out = {}
        for kind in ("ce", "kl", "bce"):
            for direction in ("sp", "po"):
                ent_master = ent[lo:hi].clone().requires_grad_(True)
                rel_master = rel.clone().requires_grad_(True)
                sh = ShardedEntityTable(model, ent_master.detach().clone(), rel_master.detach().clone(), E,
                                        backend=OracleBackend)
                masks = {"all": m_all[lo:hi], "a": m_a, "p": m_p}
                ids = s if direction == "sp" else o
                if kind == "ce":
                    rows = sh.ce_loss(direction, ids, p, o if direction == "sp" else s, ent_master, rel_master,
                                      dropout=(pe, pr), masks=masks)
                elif kind == "kl":
                    rows = sh.kl_loss(direction, ids, p, rowptr, col, ent_master, rel_master, dropout=(pe, pr), masks=masks)
                else:
                    rows = sh.bce_loss(direction, ids, p, rowptr, col, 0.25, ent_master, rel_master, dropout=(pe, pr),
                                       masks=masks)
                (rows * w).sum().backward()
                out[(kind, direction)] = (rows.detach().numpy(), ent_master.grad.numpy(), rel_master.grad.numpy())
        q.put((rank, lo, hi, out, dict(ent=ent.numpy(), rel=rel.numpy(), s=s.numpy(), p=p.numpy(), o=o.numpy(), w=w.numpy(),
                                       rowptr=rowptr.numpy(), col=col.numpy(), m_all=m_all.numpy(), m_a=m_a.numpy(),
                                       m_p=m_p.numpy(), pe=pe, pr=pr)))
    finally:
        _quiet_teardown()


@pytest.mark.parametrize("model", ["complex", "distmult"])
def test_sharded_losses_with_embedder_dropout_equal_the_unsharded_ops_on_the_same_masks(model):
    """LookupEmbedder._postprocess (kge/model/embedder/lookup_embedder.py:64-69, 102-105) applies dropout to the query
    rows, the relation rows and ALL entity rows of a training step (kge_model.py:682-725).  ShardedEntityTable's three
    losses with `dropout=`: query rows fetched in float32 from their owners, masked alike on every rank, every rank
    masking its own rows of the table; gradients through the masks, the query rows' summed over the shards and
    scattered on their owners.  Two gloo ranks with the masks handed in == the reference's op sequence on the unsharded
    tables with the same masks: per-row losses of 1vsAll CE, KvsAll KL and KvsAll BCE, both directions, the entity
    gradient (the shards' rows side by side) and the relation gradient (identical on every rank)."""
    import torch.nn.functional as F
    import torch_port as tp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_dropout_worker, args=(r, world, port, model, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    outs = []
    import time
    t0 = time.time()
    while len(outs) < world and time.time() - t0 < 180:
        if not q.empty():
            outs.append(q.get())
        elif any(pr.exitcode not in (None, 0) for pr in procs):
            break
        else:
            time.sleep(0.05)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert len(outs) == world
    outs.sort(key=lambda x: x[0])
    c = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in outs[0][4].items()}
    E, n = c["ent"].shape[0], c["s"].shape[0]
    y = torch.zeros(n, E)
    for i in range(n):
        y[i, c["col"][c["rowptr"][i]:c["rowptr"][i + 1]]] = 1.0
    k = y.sum(1)
    for kind in ("ce", "kl", "bce"):
        for direction in ("sp", "po"):
            ent_t, rel_t = c["ent"].clone().requires_grad_(True), c["rel"].clone().requires_grad_(True)
            ids = c["s"] if direction == "sp" else c["o"]
            a = ent_t[ids] * c["m_a"] / (1 - c["pe"])
            pr_ = rel_t[c["p"]] * c["m_p"] / (1 - c["pr"])
            T = ent_t * c["m_all"] / (1 - c["pe"])
            sc = tp.score_emb(model, a, pr_, T, "sp_") if direction == "sp" else tp.score_emb(model, T, pr_, a, "_po")
            if kind == "ce":
                ref = F.cross_entropy(sc, c["o"] if direction == "sp" else c["s"], reduction="none")
            elif kind == "kl":
                # KLDivWithSoftmaxKgeLoss (kge/util/loss.py:192-207) per row: labels normalised to a distribution
                lp = F.log_softmax(sc, 1)
                yn = y / k.clamp(min=1.0).view(-1, 1)
                ref = torch.where(k > 0, (torch.xlogy(yn, yn) - yn * lp).sum(1), torch.zeros(n))
            else:
                ref = F.binary_cross_entropy_with_logits(sc + 0.25, y, reduction="none").sum(1)
            (ref * c["w"]).sum().backward()
            for out in outs:
                rows, _, gr = out[3][(kind, direction)]
                np.testing.assert_allclose(rows, ref.detach().numpy(), rtol=2e-5, atol=2e-5, err_msg=f"{kind} {direction}")
                np.testing.assert_allclose(gr, rel_t.grad.numpy(), rtol=1e-4, atol=2e-5, err_msg=f"{kind} {direction}")
            ge = np.concatenate([out[3][(kind, direction)][1] for out in outs])
            np.testing.assert_allclose(ge, ent_t.grad.numpy(), rtol=1e-4, atol=2e-5, err_msg=f"{kind} {direction}")
