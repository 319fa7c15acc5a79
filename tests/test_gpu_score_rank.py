"""kge_score_rank_sp_po (scoring + rank counting in one kernel) against the two-step path it replaces:
kge_score_sp_po -> kge_rank_counts_multi (EntityRankingJob._evaluate: score_sp_po, _filter_and_rank,
_get_ranks_and_num_ties; eval_entity_ranking.py:227-313).  Integer work on identical score chains: the bar is
exact equality of every count -- raw and filtered, both directions, accumulated over entity chunks, with ties,
NaN / infinite scores, filter sets that contain the true column, columns outside the chunk and hub rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tables(eng, model, E, R, d, seed, scale=0.3, flags=0):
    g = torch.Generator().manual_seed(seed)
    ent = (torch.randn(E, d, generator=g) * scale).to(torch.bfloat16).to(DEV)
    rel = (torch.randn(R, d, generator=g) * scale).to(torch.bfloat16).to(DEV)
    return eng.Tables(model, ent, rel, 1.0, flags=flags)


def _filters(rng, n, E, K, true_col, hub_rows=()):
    """K nested filter sets per row as (begin, end, col) over one value array each: a few random columns, the
    row's true column in most of them, duplicates of other rows' columns, one hub row with thousands."""
    out, prev = [], [np.zeros(0, np.int64)] * n
    for k in range(K):
        begin, end, vals = np.zeros(n, np.int64), np.zeros(n, np.int64), []
        for i in range(n):
            cnt = int(rng.integers(0, 12)) if i not in hub_rows else min(E, 3000)
            v = rng.choice(E, size=min(E, cnt), replace=False)
            if rng.random() < 0.8:
                v = np.append(v, true_col[i])
            v = np.unique(np.concatenate([v, prev[i]])).astype(np.int64)
            if i % 7 == 3 and k == 0:
                v = np.zeros(0, np.int64)  # an unknown key: empty range
            prev[i] = v
            begin[i] = len(vals)
            vals.extend(v.tolist())
            end[i] = len(vals)
        vals = np.asarray(vals if vals else [0], np.int64)
        out.append(tuple(torch.from_numpy(x).to(DEV) for x in (begin, end, vals)))
    return out


def _two_step(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, chunks, atol, rtol):
    n, K = s.numel(), len(f_sp)
    cnt = torch.zeros(2, 2, K + 1, n, dtype=torch.int64, device=DEV)
    for lo, hi in chunks:
        sub = None if (lo == 0 and hi == T.num_ent) else torch.arange(lo, hi, device=DEV)
        sc = eng.score_sp_po(T, s, p, o, sub)
        c = hi - lo
        eng.rank_counts_multi(sc[:, :c], t_sp, f_sp, lo, o.contiguous(), atol, rtol, cnt[0, 0], cnt[0, 1])
        eng.rank_counts_multi(sc[:, c:], t_po, f_po, lo, s.contiguous(), atol, rtol, cnt[1, 0], cnt[1, 1])
    return cnt


def _fused(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, chunks, atol, rtol):
    n, K = s.numel(), len(f_sp)
    cnt = torch.zeros(2, 2, K + 1, n, dtype=torch.int64, device=DEV)
    for lo, hi in chunks:
        ok = eng.score_rank_sp_po(T, s, p, o, t_sp, t_po, f_sp, f_po, atol, rtol, cnt[0, 0], cnt[0, 1], cnt[1, 0],
                                  cnt[1, 1], lo, hi)
        assert ok, "the fused path declined a configuration it documents as supported"
    return cnt


def _true_scores(eng, T, s, p, o):
    """Elements of the score matrix: each row scored against its own target (diagonal)."""
    t_sp = eng.score_sp(T, s, p, o).diagonal().contiguous()
    t_po = eng.score_po(T, p, o, s).diagonal().contiguous()
    return t_sp, t_po


CASES = [
    # model, E, R, d, n, K, chunks (None = one chunk)
    ("complex", 14541, 237, 512, 512, 2, None),
    ("distmult", 14541, 237, 512, 512, 2, None),
    ("complex", 5000, 11, 256, 100, 2, None),
    ("distmult", 4099, 7, 256, 129, 1, None),           # ragged last tile (4099 % 64 = 3), ragged row group
    ("complex", 777, 5, 512, 3, 0, None),               # no filters, a handful of rows
    ("distmult", 64, 3, 256, 1, 2, None),               # one row, one tile
    ("complex", 9000, 13, 256, 300, 2, ((0, 4097), (4097, 9000))),  # entity chunks, accumulated
    ("distmult", 20000, 13, 512, 1000, 2, ((0, 10000), (10000, 10001), (10001, 20000))),
    ("distmult", 3000, 5, 256, 1151, 1, None),          # 9 + 9 row groups: column groups in whole groups of eight
    ("complex", 2100, 5, 256, 2500, 2, None),           # more rows than one launch holds: row blocks of 2,048
    # ONE filter set on a table whose rows of filter bits are 37 KB apart (the unused second word load of a row once
    # went through the row's offset into a buffer that has no such rows); a table beyond the L2, the deep d = 256 ring
    ("distmult", 300007, 5, 256, 512, 1, None),
]


@pytest.mark.parametrize("kernel", ["v8", "v8-split", "v4"])
@pytest.mark.parametrize("model,E,R,d,n,K,chunks", CASES)
def test_fused_counts_equal_the_two_step_counts(model, E, R, d, n, K, chunks, kernel, monkeypatch, kge_switch):
    """kernel: "v8" pairs_bf16_v8_rank_kernel (two consumer waves per SIMD: the default), "v8-split" the same with
    split queries (KGE_FLAG_SPLIT_QUERY: the parity-compliant evaluation mode; the two-step path stores the split
    scores), "v4" the round-3 epilogue of pairs_bf16_v4_kernel (KGE_V8_RANK=0: what declined launches fall back to)."""
    from kge_amd import engine as eng
    if kernel == "v4":
        kge_switch.set("V8_RANK", "0")
    rng = np.random.default_rng(E + 31 * n + K)
    T = _tables(eng, model, E, R, d, seed=E + n, flags=eng.FLAG_SPLIT_QUERY if kernel == "v8-split" else 0)
    s = torch.from_numpy(rng.integers(0, E, n)).to(DEV)
    p = torch.from_numpy(rng.integers(0, R, n)).to(DEV)
    o = torch.from_numpy(rng.integers(0, E, n)).to(DEV)
    # ties: duplicate entity rows (identical scores for every query) next to the originals
    dup = rng.integers(0, E, min(E // 2, 40))
    T.ent[dup] = T.ent[rng.integers(0, E, len(dup))]
    t_sp, t_po = _true_scores(eng, T, s, p, o)
    f_sp = _filters(rng, n, E, K, o.cpu().numpy(), hub_rows=(n // 2,))
    f_po = _filters(rng, n, E, K, s.cpu().numpy(), hub_rows=(0,))
    chunks = chunks or ((0, E),)
    for atol, rtol in ((1e-5, 1e-4), (0.05, 0.0)):  # the reference's tolerances; a band wide enough for many ties
        want = _two_step(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, chunks, atol, rtol)
        got = _fused(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, chunks, atol, rtol)
        assert torch.equal(got, want), (model, E, n, K, atol,
                                        (got != want).nonzero()[:5].tolist(), got[got != want][:5].tolist(),
                                        want[got != want][:5].tolist())
        assert int(want[:, 1].min()) >= 1  # every row is at least close to itself
    # the filter-bit buffer is all-zero again
    for buf in eng._RANK_BITS.values():
        assert int(buf.count_nonzero()) == 0


def test_fused_counts_with_nan_and_infinite_scores():
    """Rows whose true score is NaN (-> -inf), +inf or -inf, and NaN / infinite scores elsewhere: the generic
    arithmetic of the counting epilogue (taken by a wave as soon as one of its rows needs it)."""
    from kge_amd import engine as eng
    E, R, d, n = 3000, 5, 256, 160
    rng = np.random.default_rng(5)
    T = _tables(eng, "distmult", E, R, d, seed=9)
    inf = float("inf")
    T.ent[10, 0] = inf          # scores against entity 10: +-inf or NaN depending on the query's sign / zero
    T.ent[11, 3] = -inf
    T.ent[12, 5] = float("nan")
    T.ent[13, :] = 0.0
    s = torch.from_numpy(rng.integers(14, E, n)).to(DEV)
    p = torch.from_numpy(rng.integers(0, R, n)).to(DEV)
    o = torch.from_numpy(rng.integers(14, E, n)).to(DEV)
    o[:8] = torch.tensor([10, 11, 12, 13, 10, 11, 12, 13], device=DEV)  # true scores: inf / -inf / NaN / 0
    s[40:44] = torch.tensor([10, 11, 12, 13], device=DEV)               # whole rows of NaN / inf queries
    t_sp, t_po = _true_scores(eng, T, s, p, o)
    assert not bool(torch.isfinite(t_sp[:3]).any())
    f_sp = _filters(rng, n, E, 2, o.cpu().numpy())
    f_po = _filters(rng, n, E, 2, s.cpu().numpy())
    for rows in (slice(0, n), slice(48, n)):  # with and without the special rows in the first wave
        a = [x[rows] for x in (s, p, o, t_sp, t_po)]
        fs = [tuple(y[rows] if j < 2 else y for j, y in enumerate(f)) for f in f_sp]
        fp = [tuple(y[rows] if j < 2 else y for j, y in enumerate(f)) for f in f_po]
        a[3], a[4] = a[3].contiguous(), a[4].contiguous()
        fs = [(b.contiguous(), e.contiguous(), c) for b, e, c in fs]
        fp = [(b.contiguous(), e.contiguous(), c) for b, e, c in fp]
        want = _two_step(eng, T, *a, fs, fp, ((0, E),), 1e-5, 1e-4)
        got = _fused(eng, T, *a, fs, fp, ((0, E),), 1e-5, 1e-4)
        assert torch.equal(got, want), ((got != want).nonzero()[:8].tolist(), got[got != want][:8].tolist(),
                                        want[got != want][:8].tolist())


def test_fused_path_declines_what_it_does_not_cover():
    """bf16 ComplEx / DistMult at a dim the matrix-core counting kernels do not take (its store path is another bf16
    kernel without a counting epilogue; with split queries the f32 chain scores there): declined, nothing counted."""
    from kge_amd import engine as eng
    g = torch.Generator().manual_seed(1)
    ent, rel = torch.randn(500, 128, generator=g).bfloat16().to(DEV), torch.randn(4, 128, generator=g).bfloat16().to(DEV)
    s = p = o = torch.zeros(4, dtype=torch.int64, device=DEV)
    z = torch.zeros(4, device=DEV)
    for T in (eng.Tables("complex", ent, rel, 1.0),
              eng.Tables("distmult", ent, rel, 1.0, eng.FLAG_SPLIT_QUERY)):
        cnt = torch.zeros(4, 1, 4, dtype=torch.int64, device=DEV)
        assert eng.score_rank_sp_po(T, s, p, o, z, z, [], [], 1e-5, 1e-4, cnt[0], cnt[1], cnt[2], cnt[3]) is False
        assert int(cnt.abs().sum()) == 0


EXACT_CASES = [
    # model, dtype, flags, E, R, d, n, K, chunks: the counting epilogue of the exact kernels (score_pairs.hip,
    # score_pairs_f32.hip) -- float32 tables of every scorer (a LibKGE model's default precision), TransE / RotatE on
    # bf16 tables, bf16 ComplEx under KGE_FLAG_EXACT
    ("complex", torch.float32, 0, 14541, 237, 512, 512, 2, None),      # 128 x 128 tiles on the f32 matrix cores
    ("distmult", torch.float32, 0, 4099, 7, 256, 129, 1, None),        # ragged tiles in both directions
    ("complex", torch.float32, 0, 3000, 5, 128, 40, 2, None),          # n <= 64: the 64 x 64 kernel (MFMA)
    ("distmult", torch.float32, 2, 2000, 5, 100, 70, 2, None),         # KGE_FLAG_NO_MFMA, d % 8 != 0: scalar loads
    ("transe", torch.float32, 0, 5000, 11, 128, 200, 2, None),
    ("rotate", torch.float32, 0, 3001, 11, 64, 65, 2, ((0, 1500), (1500, 3001))),  # entity chunks, accumulated
    ("transe", torch.bfloat16, 0, 2500, 5, 256, 100, 1, None),
    ("complex", torch.bfloat16, 1, 3000, 5, 256, 150, 2, None),        # KGE_FLAG_EXACT
    ("distmult", torch.float32, 0, 64, 3, 64, 1, 0, None),             # one row, one tile, no filters
]


@pytest.mark.parametrize("model,dtype,flags,E,R,d,n,K,chunks", EXACT_CASES)
def test_exact_kernels_count_what_their_store_path_gives(model, dtype, flags, E, R, d, n, K, chunks):
    from kge_amd import engine as eng
    rng = np.random.default_rng(E + 31 * n + K + d)
    g = torch.Generator().manual_seed(E + n)
    ent = (torch.randn(E, d, generator=g) * 0.3).to(dtype).to(DEV)
    rel = (torch.randn(R, d // 2 if model == "rotate" else d, generator=g) * 0.3).to(dtype).to(DEV)
    T = eng.Tables(model, ent, rel, 1.0, flags)
    s = torch.from_numpy(rng.integers(0, E, n)).to(DEV)
    p = torch.from_numpy(rng.integers(0, R, n)).to(DEV)
    o = torch.from_numpy(rng.integers(0, E, n)).to(DEV)
    dup = rng.integers(0, E, min(E // 2, 40))
    T.ent[dup] = T.ent[rng.integers(0, E, len(dup))]  # ties
    T.ent[5, 0] = float("nan")                        # a NaN column (-> -inf) ...
    if n > 3:
        o[3] = 5                                      # ... that is also one row's true column
    t_sp, t_po = _true_scores(eng, T, s, p, o)
    f_sp = _filters(rng, n, E, K, o.cpu().numpy(), hub_rows=(n // 2,))
    f_po = _filters(rng, n, E, K, s.cpu().numpy(), hub_rows=(0,))
    chunks = chunks or ((0, E),)
    for atol, rtol in ((1e-5, 1e-4), (0.05, 0.0)):
        want = _two_step(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, chunks, atol, rtol)
        got = _fused(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, chunks, atol, rtol)
        assert torch.equal(got, want), (model, E, n, K, atol,
                                        (got != want).nonzero()[:5].tolist(), got[got != want][:5].tolist(),
                                        want[got != want][:5].tolist())
    for buf in eng._RANK_BITS.values():
        assert int(buf.count_nonzero()) == 0


@pytest.mark.parametrize("model,E,d,bs,chunk,dtype", [
    ("complex", 3000, 256, 100, -1, torch.bfloat16), ("distmult", 5003, 512, 77, -1, torch.bfloat16),
    ("complex", 2500, 512, 64, 999, torch.bfloat16), ("distmult", 700, 256, 300, 64, torch.bfloat16),
    ("complex", 3000, 128, 100, -1, torch.float32), ("transe", 2000, 64, 77, -1, torch.float32),
    ("rotate", 1500, 64, 64, 999, torch.float32)])
def test_evaluator_with_fused_counting_equals_the_two_step_evaluator(model, E, d, bs, chunk, dtype, monkeypatch):
    """EntityRankingEvaluator (the mirror of EntityRankingJob._evaluate) takes the fused entry for bf16 ComplEx /
    DistMult tables and for float32 tables of every scorer: per-example ranks (raw, filtered, filtered-with-test;
    both directions; ragged last batch; entity chunks) and metrics identical to the same loop over kge_score_sp_po +
    kge_rank_counts_multi."""
    from kge_amd import engine as eng
    from kge_amd.eval import EntityRankingEvaluator
    from kge_amd.synthetic import make_splits
    R = 9
    splits = make_splits(E, R, 6000, 431, 300, seed=E)
    if dtype == torch.bfloat16:
        T = _tables(eng, model, E, R, d, seed=d + E)
    else:
        g = torch.Generator().manual_seed(d + E)
        T = eng.Tables(model, (torch.randn(E, d, generator=g) * 0.3).to(DEV),
                       (torch.randn(R, d // 2 if model == "rotate" else d, generator=g) * 0.3).to(DEV), 1.0)
    calls = {"fused": 0}
    orig, orig_batch = eng.score_rank_sp_po, eng.eval_batch
    # float32 tables: by default only from 1 GiB of score matrix on
    monkeypatch.setitem(EntityRankingEvaluator.OPTIONS, "fused_exact", True)

    def counting(*a, **k):
        calls["fused"] += 1
        return orig(*a, **k)

    def counting_batch(*a, **k):
        calls["fused"] += 1
        return orig_batch(*a, **k)

    eng.score_rank_sp_po, eng.eval_batch = counting, counting_batch
    try:
        ev = EntityRankingEvaluator(T, splits, E, R, eval_split="valid", batch_size=bs, chunk_size=chunk)
        m1, r1 = ev.run(return_ranks=True)
        assert calls["fused"] > 0 and ev._fused
        # full batches of the unchunked loop replay captured hipGraphs, `ev.lanes` of them in flight on as many
        # streams (each lane's first batch is the one it captures); the same loop issued launch by launch
        full = 431 // bs
        assert ev.lanes >= 2
        assert ev.graph_batches == (max(0, full - min(ev.lanes, full)) if chunk < 0 and full >= 4 else 0)
        ev3 = EntityRankingEvaluator(T, splits, E, R, eval_split="valid", batch_size=bs, chunk_size=chunk)
        ev3.hip_graph = False
        m3, r3 = ev3.run(return_ranks=True)
        assert ev3.graph_batches == 0 and m3 == m1 and all(np.array_equal(r1[k], r3[k]) for k in r1)
        # unchunked batches go through kge_eval_batch (four launches); the same loop as separate engine calls
        ev4 = EntityRankingEvaluator(T, splits, E, R, eval_split="valid", batch_size=bs, chunk_size=chunk)
        ev4.four_launches = False
        m4, r4 = ev4.run(return_ranks=True)
        assert m4 == m1 and all(np.array_equal(r1[k], r4[k]) for k in r1)
        m1b, _ = ev.run(return_ranks=True)  # a second run captures afresh
        assert m1b == m1
        ev1 = EntityRankingEvaluator(T, splits, E, R, eval_split="valid", batch_size=bs, chunk_size=chunk)
        ev1.lanes = 1  # one captured batch at a time
        m5, r5 = ev1.run(return_ranks=True)
        assert m5 == m1 and all(np.array_equal(r1[k], r5[k]) for k in r1)
        ev2 = EntityRankingEvaluator(T, splits, E, R, eval_split="valid", batch_size=bs, chunk_size=chunk)
        ev2._fused = False
        before = calls["fused"]
        m2, r2 = ev2.run(return_ranks=True)
        assert calls["fused"] == before
    finally:
        eng.score_rank_sp_po, eng.eval_batch = orig, orig_batch
    for k in r2:
        assert np.array_equal(r1[k], r2[k]), (k, np.nonzero(r1[k] != r2[k])[0][:5])
    assert m1 == m2


def test_fused_counts_against_the_oracle():
    """Directly against the CPU oracle's restatement of _filter_and_rank / _get_ranks_and_num_ties
    (oracle.rank_counts: dense 0 / inf label subtraction, isclose, two sums -- eval_entity_ranking.py:533-596) applied
    to the score matrix the GPU writes: raw and two filtered rankings, both directions, two entity chunks."""
    import oracle as ko
    from kge_amd import engine as eng
    model, E, R, d, n = "complex", 6000, 9, 256, 200
    rng = np.random.default_rng(77)
    T = _tables(eng, model, E, R, d, seed=77)
    s = torch.from_numpy(rng.integers(0, E, n)).to(DEV)
    p = torch.from_numpy(rng.integers(0, R, n)).to(DEV)
    o = torch.from_numpy(rng.integers(0, E, n)).to(DEV)
    T.ent[rng.integers(0, E, 30)] = T.ent[rng.integers(0, E, 30)]  # exact ties
    sc = eng.score_sp_po(T, s, p, o)
    ar = torch.arange(n, device=DEV)
    t_sp, t_po = sc[ar, o].contiguous(), sc[ar, E + s].contiguous()
    f_sp = _filters(rng, n, E, 2, o.cpu().numpy(), hub_rows=(3,))
    f_po = _filters(rng, n, E, 2, s.cpu().numpy(), hub_rows=(5,))
    got = _fused(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, ((0, 2500), (2500, E)), 1e-5, 1e-4).cpu().numpy()
    scn = sc.cpu().numpy()
    for side, (blk, true, tcol, filt) in enumerate(((scn[:, :E], t_sp, o, f_sp), (scn[:, E:], t_po, s, f_po))):
        blk, true, tcol = np.ascontiguousarray(blk), true.cpu().numpy(), tcol.cpu().numpy()
        r0, t0 = ko.rank_counts(blk, true)
        assert np.array_equal(got[side, 0, 0], r0) and np.array_equal(got[side, 1, 0], t0), ("raw", side)
        for k, (beg, end, vals) in enumerate(filt):
            beg, end, vals = beg.cpu().numpy(), end.cpu().numpy(), vals.cpu().numpy()
            rp = np.concatenate([[0], np.cumsum(end - beg)]).astype(np.int64)
            col = np.concatenate([vals[b:e] for b, e in zip(beg, end)] + [np.zeros(0, np.int64)]).astype(np.int64)
            r1, t1 = ko.rank_counts(blk, true, rp, col, 0, tcol, 1e-5, 1e-4)
            assert np.array_equal(got[side, 0, k + 1], r1) and np.array_equal(got[side, 1, k + 1], t1), (k, side)


# ---- band-and-rescore (round 6) -------------------------------------------------------------------------
def _split_vs_band(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, chunks, atol, rtol):
    """(counts of the split kernel, counts with band-and-rescore, [(pairs listed, pairs dropped, pairs) per chunk])."""
    n, K = s.numel(), len(f_sp)
    want = _fused(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, chunks, atol, rtol)
    got = torch.zeros(2, 2, K + 1, n, dtype=torch.int64, device=DEV)
    stats = []
    for lo, hi in chunks:
        band = eng.RankBand(T, n, lo, hi)
        ok = eng.score_rank_sp_po(T, s, p, o, t_sp, t_po, f_sp, f_po, atol, rtol, got[0, 0], got[0, 1], got[1, 0],
                                  got[1, 1], lo, hi, band=band)
        assert ok
        stats.append(band.status() + (band.pairs_of(n),))
        # every list is empty again (header word 0 of every 4096-byte list)
        assert int(band.list.view(torch.int32).view(-1, 1024)[:, 0].abs().sum()) == 0
    return want, got, stats


def _planted(eng, T, rng, E, R, n, kth):
    """(s, p, o) whose true object is the kth best entity of (s, p) -- which puts the same score into the tail of
    (p, o)'s row: a trained model's evaluation triples."""
    s = torch.from_numpy(rng.integers(0, E, n)).to(DEV)
    p = torch.from_numpy(rng.integers(0, R, n)).to(DEV)
    o = torch.empty(n, dtype=torch.int64, device=DEV)
    for i0 in range(0, n, 128):
        o[i0:i0 + 128] = eng.score_sp(T, s[i0:i0 + 128], p[i0:i0 + 128]).topk(kth, dim=1).indices[:, -1]
    return s, p, o


@pytest.mark.parametrize("model,E,R,d,n,K,chunks", CASES)
def test_band_and_rescore_counts_equal_the_split_kernels(model, E, R, d, n, K, chunks):
    """kge_score_rank_sp_po_band against kge_score_rank_sp_po under KGE_FLAG_SPLIT_QUERY on the shapes of the table
    above (ragged tiles and row groups, chunks, hub rows, one / two / no filter sets, duplicate entity rows = exact
    ties), on triples that rank high (the k-th best entity of (s, p): what the band is for): every count equal, no pair
    dropped, every list empty afterwards.  With the wide tie band (0.05) of the second pass -- and on the small tables,
    where the k-th best of a few hundred entities is not far out -- the lists overflow: then the dropped pairs must be
    REPORTED (the caller's signal to count the batch again), and the counts must not exceed the split kernel's."""
    from kge_amd import engine as eng
    rng = np.random.default_rng(E + 31 * n + K)
    T = _tables(eng, model, E, R, d, seed=E + n, flags=eng.FLAG_SPLIT_QUERY)
    s, p, o = _planted(eng, T, rng, E, R, n, max(2, E // 10000))
    dup = rng.integers(0, E, min(E // 2, 40))
    T.ent[dup] = T.ent[rng.integers(0, E, len(dup))]
    T.ent[rng.integers(0, E, min(4, n))] = T.ent[o[:4]]       # exact ties with true rows
    t_sp, t_po = _true_scores(eng, T, s, p, o)
    f_sp = _filters(rng, n, E, K, o.cpu().numpy(), hub_rows=(n // 2,))
    f_po = _filters(rng, n, E, K, s.cpu().numpy(), hub_rows=(0,))
    chunks = chunks or ((0, E),)
    complete, seen = 0, []
    for atol, rtol in ((1e-5, 1e-4), (0.05, 0.0)):
        want, got, stats = _split_vs_band(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, chunks, atol, rtol)
        seen.append(stats)
        if all(st[1] == 0 for st in stats):
            complete += 1
            assert all(st[0] > 0 for st in stats), stats   # (the true column itself is always inside its band)
            assert torch.equal(got, want), (model, E, n, K, atol, stats, (got != want).nonzero()[:5].tolist(),
                                            got[got != want][:5].tolist(), want[got != want][:5].tolist())
        else:
            assert bool((got <= want).all()), (model, E, n, K, atol, stats)
    assert complete >= (1 if E >= 2000 else 0), (model, E, n, K, seen)
    for buf in eng._RANK_BITS.values():
        assert int(buf.count_nonzero()) == 0


@pytest.mark.parametrize("model,E,d,n", [("complex", 60000, 256, 512), ("distmult", 14541, 512, 512),
                                         ("complex", 300007, 256, 300), ("distmult", 100003, 512, 1100)])
def test_band_and_rescore_on_planted_true_scores_lists_few_pairs(model, E, d, n):
    """The band's regime at size: ~5e-5 of the pairs listed, none dropped, counts == the split kernel's."""
    from kge_amd import engine as eng
    R, K = 7, 2
    rng = np.random.default_rng(E + d)
    T = _tables(eng, model, E, R, d, seed=E + d, flags=eng.FLAG_SPLIT_QUERY)
    s, p, o = _planted(eng, T, rng, E, R, n, max(2, E // 10000))
    T.ent[rng.integers(0, E, 8)] = T.ent[o[:8]]       # exact ties with true rows
    t_sp, t_po = _true_scores(eng, T, s, p, o)
    f_sp = _filters(rng, n, E, K, o.cpu().numpy(), hub_rows=(n // 2,))
    f_po = _filters(rng, n, E, K, s.cpu().numpy(), hub_rows=(0,))
    want, got, stats = _split_vs_band(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, ((0, E),), 1e-5, 1e-4)
    listed, dropped, pairs = stats[0]
    assert dropped == 0 and 2 * n <= listed < 5e-4 * pairs, stats
    assert torch.equal(got, want), ((got != want).nonzero()[:5].tolist(), got[got != want][:5].tolist(),
                                    want[got != want][:5].tolist())
    # a band without split queries is an argument error
    T1 = _tables(eng, model, E, R, d, seed=E + d)
    cnt = torch.zeros(2, 2, K + 1, n, dtype=torch.int64, device=DEV)
    with pytest.raises(Exception):
        eng.score_rank_sp_po(T1, s, p, o, t_sp, t_po, f_sp, f_po, 1e-5, 1e-4, cnt[0, 0], cnt[0, 1], cnt[1, 0], cnt[1, 1],
                             band=eng.RankBand(T1, n))
    assert int(cnt.abs().sum()) == 0


def test_band_on_random_triples_reports_the_pairs_it_drops():
    """Random triples put the true score into the bulk of its row: 1.5 % of the pairs are inside the band, the waves'
    lists (255 pairs each) overflow, and the status words say so -- sticky over calls; lists and filter bits are clean
    afterwards and a following call on triples that rank high is complete and exact."""
    from kge_amd import engine as eng
    E, R, d, n, K = 14541, 11, 512, 512, 2
    rng = np.random.default_rng(3)
    T = _tables(eng, "complex", E, R, d, seed=3, flags=eng.FLAG_SPLIT_QUERY)
    s, p, o = (torch.from_numpy(rng.integers(0, hi, n)).to(DEV) for hi in (E, R, E))
    t_sp, t_po = _true_scores(eng, T, s, p, o)
    f_sp = _filters(rng, n, E, K, o.cpu().numpy())
    f_po = _filters(rng, n, E, K, s.cpu().numpy())
    band = eng.RankBand(T, n)
    cnt = torch.zeros(2, 2, K + 1, n, dtype=torch.int64, device=DEV)
    seen = []
    for call in (1, 2):
        assert eng.score_rank_sp_po(T, s, p, o, t_sp, t_po, f_sp, f_po, 1e-5, 1e-4, cnt[0, 0], cnt[0, 1], cnt[1, 0],
                                    cnt[1, 1], band=band)
        seen.append(band.status())
    assert seen[0][1] > 0 and seen[1] == (2 * seen[0][0], 2 * seen[0][1]), seen
    assert seen[0][0] > 1e-3 * band.pairs_of(n)
    assert int(band.list.view(torch.int32).view(-1, 1024)[:, 0].abs().sum()) == 0
    for buf in eng._RANK_BITS.values():
        assert int(buf.count_nonzero()) == 0
    s, p, o = _planted(eng, T, rng, E, R, n, 2)
    t_sp, t_po = _true_scores(eng, T, s, p, o)
    want, got, stats = _split_vs_band(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, ((0, E),), 1e-5, 1e-4)
    assert stats[0][1] == 0 and torch.equal(got, want)


def test_band_and_rescore_with_nan_and_infinite_scores():
    """Rows without a finite band (true score NaN / +-inf) list every pair of their wave -- dropped pairs, reported --;
    finite rows of OTHER waves are unaffected; a table with an infinite entry has no finite row-norm bound at all."""
    from kge_amd import engine as eng
    E, R, d, n = 3000, 5, 256, 160
    rng = np.random.default_rng(5)
    T = _tables(eng, "distmult", E, R, d, seed=9, flags=eng.FLAG_SPLIT_QUERY)
    s, p, o = _planted(eng, T, rng, E, R, n, 2)
    t_sp, t_po = _true_scores(eng, T, s, p, o)
    f_sp = _filters(rng, n, E, 2, o.cpu().numpy())
    f_po = _filters(rng, n, E, 2, s.cpu().numpy())
    want, got, stats = _split_vs_band(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, ((0, E),), 1e-5, 1e-4)
    assert stats[0][1] == 0 and torch.equal(got, want)
    t_bad = t_sp.clone()
    t_bad[:3] = torch.tensor([float("inf"), float("-inf"), float("nan")], device=DEV)   # rows of the first wave
    want, got, stats = _split_vs_band(eng, T, s, p, o, t_bad, t_po, f_sp, f_po, ((0, E),), 1e-5, 1e-4)
    assert stats[0][1] > 0                                   # the first wave's lists overflowed: reported
    assert torch.equal(got[:, :, :, 32:], want[:, :, :, 32:])   # the other waves' rows are complete and exact
    assert torch.equal(got[1], want[1])                      # ... and the whole po side
    T.ent[10, 0] = float("inf")
    band = eng.RankBand(T, n)
    assert not bool(torch.isfinite(band.tmax).all())
    want, got, stats = _split_vs_band(eng, T, s, p, o, t_sp, t_po, f_sp, f_po, ((0, E),), 1e-5, 1e-4)
    assert stats[0][1] > 0 and bool((got <= want).all())


@pytest.mark.parametrize("model,E,d,bs", [("complex", 60000, 256, 256), ("distmult", 30000, 512, 200)])
def test_evaluator_takes_band_and_rescore_on_tables_that_rank_their_triples_high(model, E, d, bs, monkeypatch):
    """EntityRankingEvaluator with split queries (the parity-compliant setting) and band_rescore "auto": on an
    evaluation split whose triples score in the tail of their rows (a trained model) the probe batch lists ~5e-5 of its
    pairs, the run counts through kge_eval_batch_band (lanes, captured batches, ragged last batch), and per-example
    ranks and metrics are those of the split kernel.  On random triples the probe lists 1.5 % and the evaluator keeps
    the split kernel; forced on (band_rescore=True) such a run drops pairs, notices at its end, and runs again on the
    split kernel -- the same result."""
    from kge_amd import engine as eng
    from kge_amd.eval import EntityRankingEvaluator
    from kge_amd.synthetic import make_splits
    R = 9
    rng = np.random.default_rng(E)
    T = _tables(eng, model, E, R, d, seed=d + E, flags=eng.FLAG_SPLIT_QUERY)
    splits = make_splits(E, R, 6000, 8 * bs + 37, 300, seed=E)
    random_valid = splits["valid"].copy()
    s, p, o = _planted(eng, T, rng, E, R, len(splits["valid"]), max(2, E // 10000))
    splits["valid"] = torch.stack([s, p, o], 1).cpu().numpy()
    calls = {"band": 0, "plain": 0}
    orig = eng.eval_batch

    def counting(*a, **k):
        calls["band" if k.get("band") is not None else "plain"] += 1
        return orig(*a, **k)
    monkeypatch.setattr(eng, "eval_batch", counting)
    monkeypatch.setattr(EntityRankingEvaluator, "BAND_MIN_ENTITIES", 1000)   # (auto takes effect from 100,000 entities on)
    ev = EntityRankingEvaluator(T, splits, E, R, eval_split="valid", batch_size=bs)
    m1, r1 = ev.run(return_ranks=True)
    assert ev.band_runs == 1 and not ev._band_off and calls["band"] >= 3, (ev.band_listed, calls)
    listed, pairs = ev.band_listed
    assert 0 < listed <= ev.BAND_MAX_LISTED * pairs
    m1b, r1b = ev.run(return_ranks=True)   # a second run: held graphs, refreshed bands
    assert ev.band_runs == 2 and m1b == m1 and all(np.array_equal(r1[k], r1b[k]) for k in r1)
    ref = EntityRankingEvaluator(T, splits, E, R, eval_split="valid", batch_size=bs, band_rescore=False)
    before = calls["band"]
    m2, r2 = ref.run(return_ranks=True)
    assert calls["band"] == before and ref.band_runs == 0
    assert m1 == m2
    for k in r2:
        assert np.array_equal(r1[k], r2[k]), (k, np.nonzero(r1[k] != r2[k])[0][:5])
    assert float(m1["mean_reciprocal_rank"]) > 0.1   # (the planted triples do rank high)
    # random triples: the probe sees a bulk true score in every row
    splits["valid"] = random_valid
    ev2 = EntityRankingEvaluator(T, splits, E, R, eval_split="valid", batch_size=bs)
    m5 = ev2.run()
    assert ev2.band_runs == 0 and ev2.band_listed[0] > ev2.BAND_MAX_LISTED * ev2.band_listed[1]
    # ... forced on: pairs are dropped, the run notices and repeats itself on the split kernel
    ev3 = EntityRankingEvaluator(T, splits, E, R, eval_split="valid", batch_size=bs, band_rescore=True)
    m3, r3 = ev3.run(return_ranks=True)
    ref3 = EntityRankingEvaluator(T, splits, E, R, eval_split="valid", batch_size=bs, band_rescore=False)
    m4, r4 = ref3.run(return_ranks=True)
    assert ev3._band_off and ev3.band_runs == 0 and m3 == m4 == m5 and all(np.array_equal(r3[k], r4[k]) for k in r4)
