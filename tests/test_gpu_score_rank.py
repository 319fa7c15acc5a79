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
    monkeypatch.setenv("KGE_EVAL_FUSED_EXACT", "1")  # float32 tables: by default only from 1 GiB of score matrix on

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
