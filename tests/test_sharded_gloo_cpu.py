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
        slab = sh.score_sp(tb[:, 0], tb[:, 1])
        tv, ti = sh.topk(slab, 5)
        if rank == 0:
            q.put((out, tv.numpy(), ti.numpy(), ent, rel, splits, batch))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model", ["complex", "transe", "rotate"])
def test_two_shards_equal_unsharded(model):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, model, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, tv, ti, ent, rel, splits, batch = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
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
    order = np.argsort(-sp, axis=1, kind="stable")[:, :5]
    assert np.array_equal(np.take_along_axis(sp, order, 1), tv)
    assert np.array_equal(np.sort(ti, 1), np.sort(order, 1)) or np.allclose(np.take_along_axis(sp, ti, 1), tv)
