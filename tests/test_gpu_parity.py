"""GPU parity tests proper: the HIP path (through the C ABI, via kge_amd.engine) against the
CPU oracle on the golden inputs and on seeded random inputs.

Bars:
  * f32 arithmetic paths -- every scorer, spo / sp_ / _po / sp_po / negatives, f32 and bf16
    tables (bf16 widened exactly), and the FLAG_EXACT twin of the bf16 MFMA kernel:
    BIT-EXACT against the oracle (values compared with ==; +0 and -0 compare equal).
  * bf16 MFMA kernel (ComplEx/DistMult, d % 64 == 0): the MFMA's internal summation order
    is unspecified -> reference tolerance atol=1e-5*scale, rtol=1e-4 against the oracle.
  * everything against the reference's golden outputs: reference tolerance.
  * rank counts: integer, exact.
"""
import glob
import os

import numpy as np
import pytest
import torch

import oracle as ko
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

SCORE_FILES = sorted(glob.glob(os.path.join(GOLDEN, "scores_*.npz")))
IDS = [os.path.basename(p)[7:-4] for p in SCORE_FILES]
DEV = "cuda:0"


@pytest.fixture(scope="module")
def eng():
    from kge_amd import engine
    return engine


def _np(x):
    return x.detach().cpu().numpy()


def _eq(name, got, want):
    """exact comparison with a useful failure report"""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    same = (got == want) | (np.isnan(got) & np.isnan(want))
    if not same.all():
        bad = np.argwhere(~same)
        diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
        raise AssertionError(
            f"{name}: {len(bad)}/{got.size} elements differ; max|diff|={np.nanmax(diff):.3e}; "
            f"first bad idx={bad[:5].tolist()} got={got[tuple(bad[0])]!r} want={want[tuple(bad[0])]!r}")


def _close(name, got, want, atol=1e-5, rtol=1e-4):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    fin = np.isfinite(want)
    scale = max(1.0, float(np.sqrt(np.mean(np.square(want[fin]))))) if fin.any() else 1.0
    err = np.abs(got - want)
    tol = atol * scale + rtol * np.abs(want)
    if not (err <= tol).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{name}: max excess at {i}: got={got[i]} want={want[i]} "
                             f"err={err[i]:.3e} tol={tol[i]:.3e} (scale={scale:.3g}); "
                             f"{int((err > tol).sum())}/{err.size} out of tolerance")


def _gpu_tables(eng, model, ent, rel, l_norm, bf16=False, flags=0):
    e, r = torch.from_numpy(np.ascontiguousarray(ent)), torch.from_numpy(np.ascontiguousarray(rel))
    if bf16:
        e, r = e.to(torch.bfloat16), r.to(torch.bfloat16)
    return eng.Tables(model, e.to(DEV), r.to(DEV), l_norm, flags)


def _oracle_tables(model, ent, rel, l_norm, bf16=False):
    if bf16:
        return ko.Tables(model, ko.f32_to_bf16(ent), ko.f32_to_bf16(rel), l_norm)
    return ko.Tables(model, ent, rel, l_norm)


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def test_single_hip_runtime_and_library_loaded(eng):
    from kge_amd import _lib
    assert _lib.lib().kge_device_count() >= 1
    maps = open("/proc/self/maps").read()
    hips = {l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l}
    assert len(hips) == 1, hips
    blas = {l.split()[-1] for l in maps.splitlines() if "librocblas" in l}
    assert len(blas) <= 1, blas  # (torch's own; libkge_amd.so links no BLAS: its gradient products are hand-written)
    assert "libkge_amd.so" in maps
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
@pytest.mark.parametrize("path", SCORE_FILES, ids=IDS)
def test_golden_inputs_exact_vs_oracle(eng, path, bf16):
    g = np.load(path)
    model, l_norm = str(g["model"]), float(g["l_norm"])
    T = _gpu_tables(eng, model, g["ent"], g["rel"], l_norm, bf16, flags=eng.FLAG_EXACT)
    O = _oracle_tables(model, g["ent"], g["rel"], l_norm, bf16)
    s, p, o, sub, neg = g["s"], g["p"], g["o"], g["sub"], g["neg"]
    ts, tp, to, tsub, tneg = _t(s), _t(p), _t(o), _t(sub), _t(neg)
    _eq("spo", _np(eng.score_spo(T, ts, tp, to)), ko.score_spo(O, s, p, o))
    _eq("sp", _np(eng.score_sp(T, ts, tp)), ko.score_sp(O, s, p))
    _eq("po", _np(eng.score_po(T, tp, to)), ko.score_po(O, p, o))
    _eq("sp_sub", _np(eng.score_sp(T, ts, tp, tsub)), ko.score_sp(O, s, p, sub))
    _eq("po_sub", _np(eng.score_po(T, tp, to, tsub)), ko.score_po(O, p, o, sub))
    _eq("sp_po_sub", _np(eng.score_sp_po(T, ts, tp, to, tsub)), ko.score_sp_po(O, s, p, o, sub))
    _eq("sp_po_all", _np(eng.score_sp_po(T, ts, tp, to)), ko.score_sp_po(O, s, p, o))
    _eq("neg_s", _np(eng.score_neg(T, ts, tp, to, 0, tneg)), ko.score_neg(O, s, p, o, 0, neg))
    _eq("neg_o", _np(eng.score_neg(T, ts, tp, to, 2, tneg)), ko.score_neg(O, s, p, o, 2, neg))


@pytest.mark.parametrize("path", SCORE_FILES, ids=IDS)
def test_golden_outputs_of_the_reference(eng, path):
    """HIP (f32 tables) against the outputs of the live reference, reference tolerance."""
    g = np.load(path)
    T = _gpu_tables(eng, str(g["model"]), g["ent"], g["rel"], float(g["l_norm"]))
    ts, tp, to, tsub, tneg = (_t(g[k]) for k in ("s", "p", "o", "sub", "neg"))
    _close("spo", _np(eng.score_spo(T, ts, tp, to)), g["spo"])
    _close("sp", _np(eng.score_sp(T, ts, tp)), g["sp"])
    _close("po", _np(eng.score_po(T, tp, to)), g["po"])
    _close("sp_sub", _np(eng.score_sp(T, ts, tp, tsub)), g["sp_sub"])
    _close("po_sub", _np(eng.score_po(T, tp, to, tsub)), g["po_sub"])
    _close("sp_po_sub", _np(eng.score_sp_po(T, ts, tp, to, tsub)), g["sp_po_sub"])
    _close("sp_po_all", _np(eng.score_sp_po(T, ts, tp, to)), g["sp_po_all"])
    _close("neg_s", _np(eng.score_neg(T, ts, tp, to, 0, tneg)), g["neg_s"])
    _close("neg_o", _np(eng.score_neg(T, ts, tp, to, 2, tneg)), g["neg_o"])


@pytest.mark.parametrize("model", ["complex", "distmult"])
def test_f32_mfma_equals_valu_chain(eng, model):
    """v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf chain: same bits as the VALU twin."""
    rng = np.random.default_rng(3)
    E, R, d, n = 333, 7, 96, 70
    ent = rng.standard_normal((E, d)).astype(np.float32)
    rel = rng.standard_normal((R, d)).astype(np.float32)
    s, p = rng.integers(0, E, n), rng.integers(0, R, n)
    T = _gpu_tables(eng, model, ent, rel, 1.0)
    a = _np(eng.score_sp(T, _t(s), _t(p)))
    b = _np(eng.score_sp(T, _t(s), _t(p), flags=eng.FLAG_NO_MFMA))
    _eq("mfma vs valu", a, b)
    _eq("mfma vs oracle", a, ko.score_sp(ko.Tables(model, ent, rel, 1.0), s, p))


RANDOM_CASES = [
    # model, l_norm, E, R, d, n, m_subset
    ("complex", 1.0, 1000, 11, 64, 37, 130),
    ("complex", 1.0, 700, 5, 100, 129, 65),      # d % 8 != 0 -> scalar staging
    ("complex", 1.0, 900, 9, 256, 200, 257),
    ("distmult", 1.0, 1000, 11, 100, 64, 64),
    ("distmult", 1.0, 515, 3, 37, 5, 1),         # odd d, single target
    ("distmult", 1.0, 800, 6, 512, 70, 129),
    ("transe", 1.0, 777, 4, 128, 65, 63),
    ("transe", 2.0, 600, 4, 50, 33, 200),
    ("transe", 1.0, 300, 4, 1030, 3, 17),        # d > 512: lanes own several chunks
    ("rotate", 1.0, 640, 5, 128, 66, 64),
    ("rotate", 2.0, 500, 5, 60, 20, 31),
    ("rotate", 1.0, 1100, 11, 512, 40, 100),
]


@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", RANDOM_CASES, ids=[f"{c[0]}-p{c[1]:g}-E{c[2]}-d{c[4]}-n{c[5]}" for c in RANDOM_CASES])
def test_random_shapes_exact_vs_oracle(eng, case, bf16):
    model, l_norm, E, R, d, n, msub = case
    rng = np.random.default_rng(E * 7 + d * 3 + n)
    dr = d // 2 if model == "rotate" else d
    ent = rng.standard_normal((E, d)).astype(np.float32)
    rel = (rng.uniform(-np.pi, np.pi, (R, dr)) if model == "rotate"
           else rng.standard_normal((R, dr))).astype(np.float32)
    tri = np.stack([rng.integers(0, E, n), rng.integers(0, R, n), rng.integers(0, E, n)], 1)
    sub = rng.permutation(E)[:msub]
    neg = rng.integers(0, E, (n, 9))
    T = _gpu_tables(eng, model, ent, rel, l_norm, bf16, flags=eng.FLAG_EXACT)
    O = _oracle_tables(model, ent, rel, l_norm, bf16)
    ttri = _t(tri)                       # int64 [n,3]; columns are stride-3 views
    ts, tp, to = ttri[:, 0], ttri[:, 1], ttri[:, 2]
    s, p, o = tri[:, 0], tri[:, 1], tri[:, 2]
    _eq("spo", _np(eng.score_spo(T, ts, tp, to)), ko.score_spo(O, s, p, o))
    _eq("sp_all", _np(eng.score_sp(T, ts, tp)), ko.score_sp(O, s, p))
    _eq("po_sub", _np(eng.score_po(T, tp, to, _t(sub))), ko.score_po(O, p, o, sub))
    tri32 = ttri.int()
    _eq("sp_po_sub_i32", _np(eng.score_sp_po(T, tri32[:, 0], tri32[:, 1], tri32[:, 2], _t(sub).int())),
        ko.score_sp_po(O, s, p, o, sub))
    _eq("neg_o", _np(eng.score_neg(T, ts, tp, to, 2, _t(neg))), ko.score_neg(O, s, p, o, 2, neg))
    _eq("neg_s_i32", _np(eng.score_neg(T, ts, tp, to, 0, _t(neg).int())), ko.score_neg(O, s, p, o, 0, neg))


@pytest.mark.parametrize("model,d", [("complex", 128), ("complex", 512), ("distmult", 256), ("distmult", 64)])
def test_bf16_mfma_kernel_vs_oracle(eng, model, d):
    """The headline kernel (bf16 tables, bf16 matrix cores) against the oracle's bf16
    semantics (q rounded to bf16, exact products, f32 accumulation), reference tolerance;
    ragged n and m exercise the tile tails."""
    rng = np.random.default_rng(d)
    E, R, n = 1000 + 37, 13, 200 + 3
    ent = rng.standard_normal((E, d)).astype(np.float32)
    rel = rng.standard_normal((R, d)).astype(np.float32)
    s, p, o = rng.integers(0, E, n), rng.integers(0, R, n), rng.integers(0, E, n)
    sub = rng.permutation(E)[:300]
    T = _gpu_tables(eng, model, ent, rel, 1.0, bf16=True)
    O = _oracle_tables(model, ent, rel, 1.0, bf16=True)
    ts, tp, to = _t(s), _t(p), _t(o)
    _close("sp_all", _np(eng.score_sp(T, ts, tp)), ko.score_sp(O, s, p))
    _close("po_all", _np(eng.score_po(T, tp, to)), ko.score_po(O, p, o))
    _close("sp_po_sub", _np(eng.score_sp_po(T, ts, tp, to, _t(sub))), ko.score_sp_po(O, s, p, o, sub))
    _close("sp_sub_i32", _np(eng.score_sp(T, ts.int(), tp.int(), _t(sub).int())), ko.score_sp(O, s, p, sub))
    # default = cooperative query build through the scratch buffer; without a buffer every
    # workgroup builds its own copy; v2 = 32-target tiles (with buffer: separate builder
    # kernel).  Identical semantics and K order -> the same bits everywhere.
    Tn = _gpu_tables(eng, model, ent, rel, 1.0, bf16=True)
    Tn.use_workspace = False
    ref_sp, ref_po = _np(eng.score_sp(Tn, ts, tp)), _np(eng.score_po(Tn, tp, to, _t(sub).int()))
    _close("sp_all (no workspace)", ref_sp, ko.score_sp(O, s, p))
    _eq("coop == fused", _np(eng.score_sp(T, ts, tp)), ref_sp)
    _eq("coop == fused (po, subset i32)", _np(eng.score_po(T, tp, to, _t(sub).int())), ref_po)
    _eq("v3 coop == fused", _np(eng.score_sp(T, ts, tp, flags=eng.FLAG_BF16_V3)), ref_sp)
    _eq("v3 coop == fused (po, subset i32)", _np(eng.score_po(T, tp, to, _t(sub).int(), flags=eng.FLAG_BF16_V3)), ref_po)
    _eq("16 CUs left free == default", _np(eng.score_sp(T, ts, tp, flags=eng.reserve_cus(16))), ref_sp)
    _eq("v3 fused == v3 coop", _np(eng.score_sp(Tn, ts, tp, flags=eng.FLAG_BF16_V3)), ref_sp)
    _eq("v3 fused == v3 coop (po, subset i32)", _np(eng.score_po(Tn, tp, to, _t(sub).int(), flags=eng.FLAG_BF16_V3)), ref_po)
    # the order-specified f32 chain computes the same semantics
    _eq("exact twin", _np(eng.score_sp(T, ts, tp, flags=eng.FLAG_EXACT)), ko.score_sp(O, s, p))


def test_bf16_mfma_kernel_small_and_single_row(eng):
    """n = 1 and n < 32, m < 32: every tail path of the row-persistent kernel."""
    rng = np.random.default_rng(5)
    E, R, d = 70, 3, 512
    ent = rng.standard_normal((E, d)).astype(np.float32)
    rel = rng.standard_normal((R, d)).astype(np.float32)
    T = _gpu_tables(eng, "complex", ent, rel, 1.0, bf16=True)
    O = _oracle_tables("complex", ent, rel, 1.0, bf16=True)
    for n, msub in ((1, None), (5, 7), (33, 1)):
        s, p = rng.integers(0, E, n), rng.integers(0, R, n)
        sub = None if msub is None else rng.permutation(E)[:msub]
        got = _np(eng.score_sp(T, _t(s), _t(p), None if sub is None else _t(sub)))
        _close(f"n={n}", got, ko.score_sp(O, s, p, sub))


@pytest.mark.parametrize("d,n,E", [(128, 520, 3000), (512, 300, 9000), (256, 1100, 2500)])
def test_bf16_cooperative_build_repeated_calls(eng, d, n, E):
    """Cooperative query build: several row groups, builder shares of different sizes, the
    same scratch buffer reused by consecutive calls with different queries (a stale fragment
    or flag from the previous call would show), against the no-workspace path (same bits)."""
    rng = np.random.default_rng(d + n)
    R = 11
    ent = rng.standard_normal((E, d)).astype(np.float32)
    rel = rng.standard_normal((R, d)).astype(np.float32)
    T = _gpu_tables(eng, "complex", ent, rel, 1.0, bf16=True)
    Tn = _gpu_tables(eng, "complex", ent, rel, 1.0, bf16=True)
    Tn.use_workspace = False
    outs, refs = [], []
    for it in range(6):  # back to back, no host sync in between
        s, p = _t(rng.integers(0, E, n)), _t(rng.integers(0, R, n))
        if it % 2:
            outs.append(eng.score_po(T, p, s))
            refs.append(eng.score_po(Tn, p, s))
        else:
            outs.append(eng.score_sp(T, s, p))
            refs.append(eng.score_sp(Tn, s, p))
    for it, (a, b) in enumerate(zip(outs, refs)):
        _eq(f"call {it}", _np(a), _np(b))
    # score_sp_po = one two-sided launch (sp_ row groups, then _po row groups into the second
    # column block); all entities and a ragged subset
    s, p, o = (_t(rng.integers(0, hi, n)) for hi in (E, R, E))
    sub = _t(rng.permutation(E)[: 64 * 7 + 5])
    for ss in (None, sub):
        both = _np(eng.score_sp_po(T, s, p, o, ss))
        _eq(f"two-sided sp block (subset={ss is not None})", both[:, : both.shape[1] // 2], _np(eng.score_sp(Tn, s, p, ss)))
        _eq(f"two-sided po block (subset={ss is not None})", both[:, both.shape[1] // 2:], _np(eng.score_po(Tn, p, o, ss)))
    O = _oracle_tables("complex", ent, rel, 1.0, bf16=True)
    s, p = rng.integers(0, E, 64), rng.integers(0, R, 64)
    _close("vs oracle", _np(eng.score_sp(T, _t(s), _t(p))), ko.score_sp(O, s, p))


def test_bf16_under_graph_capture(eng):
    """Under hipGraph capture the kernel arguments (the hand-off epoch included) are frozen:
    replays with new queries must still give the right scores (every consumer clears its flag
    line after reading it), back to back and with eager calls on the same scratch buffer in
    between."""
    rng = np.random.default_rng(3)
    E, R, d, n = 2000, 7, 256, 256
    ent = rng.standard_normal((E, d)).astype(np.float32)
    rel = rng.standard_normal((R, d)).astype(np.float32)
    T = _gpu_tables(eng, "distmult", ent, rel, 1.0, bf16=True)
    Tn = _gpu_tables(eng, "distmult", ent, rel, 1.0, bf16=True)
    Tn.use_workspace = False
    s, p = _t(rng.integers(0, E, n)), _t(rng.integers(0, R, n))
    eng.score_sp(T, s, p)  # warm-up outside capture (scratch buffer, module load)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = eng.score_sp(T, s, p)
    for it in range(3):
        s.copy_(_t(rng.integers(0, E, n)))
        p.copy_(_t(rng.integers(0, R, n)))
        g.replay()
        torch.cuda.synchronize()
        _eq(f"replay {it}", _np(out), _np(eng.score_sp(Tn, s, p)))
    # replays back to back (no host sync), inputs changed by stream-ordered copies in between
    news = [(_t(rng.integers(0, E, n)), _t(rng.integers(0, R, n))) for _ in range(4)]
    outs = []
    for s2, p2 in news:
        s.copy_(s2)
        p.copy_(p2)
        g.replay()
        outs.append(out.clone())
    torch.cuda.synchronize()
    for it, ((s2, p2), got) in enumerate(zip(news, outs)):
        _eq(f"back-to-back replay {it}", _np(got), _np(eng.score_sp(Tn, s2, p2)))


@pytest.mark.parametrize("model", ["complex", "distmult", "transe", "rotate"])
def test_score_emb_equals_index_level(eng, model):
    """RelationalScorer.score_emb on dense embeddings == the fused-gather entry points."""
    rng = np.random.default_rng(11)
    E, R, d, n = 300, 5, 64, 33
    dr = d // 2 if model == "rotate" else d
    ent = torch.from_numpy(rng.standard_normal((E, d)).astype(np.float32)).to(DEV)
    rel = torch.from_numpy(rng.standard_normal((R, dr)).astype(np.float32)).to(DEV)
    s, p, o = (torch.from_numpy(rng.integers(0, hi, n)).to(DEV) for hi in (E, R, E))
    T = eng.Tables(model, ent, rel, 1.0)
    se, pe, oe = ent[s], rel[p], ent[o]
    _eq("spo", _np(eng.score_emb(model, se, pe, oe, "spo")).reshape(-1), _np(eng.score_spo(T, s, p, o)))
    _eq("sp_", _np(eng.score_emb(model, se, pe, ent, "sp_")), _np(eng.score_sp(T, s, p)))
    _eq("_po", _np(eng.score_emb(model, ent, pe, oe, "_po")), _np(eng.score_po(T, p, o)))
    _eq("sp_po (dense rows)", _np(eng.score_emb_sp_po(model, se, pe, oe, ent)), _np(eng.score_sp_po(T, s, p, o)))
    with pytest.raises(ValueError):
        eng.score_emb(model, se, pe, oe, "s_x")
    if model in ("complex", "distmult"):  # the two-sided launch of the bf16 matrix-core kernel
        d5 = 512
        ent5 = torch.from_numpy(rng.standard_normal((E, d5)).astype(np.float32)).bfloat16().to(DEV)
        rel5 = torch.from_numpy(rng.standard_normal((R, d5)).astype(np.float32)).bfloat16().to(DEV)
        T5 = eng.Tables(model, ent5, rel5, 1.0)
        _eq("bf16 sp_po (dense rows, strided)", _np(eng.score_emb_sp_po(
            model, torch.cat([ent5[s], ent5[o]], 1)[:, :d5], rel5[p], torch.cat([ent5[s], ent5[o]], 1)[:, d5:], ent5)),
            _np(eng.score_sp_po(T5, s, p, o)))


def test_empty_inputs(eng):
    ent = torch.randn(10, 16, device=DEV)
    rel = torch.randn(3, 16, device=DEV)
    T = eng.Tables("distmult", ent, rel)
    z = torch.zeros(0, dtype=torch.long, device=DEV)
    assert eng.score_spo(T, z, z, z).shape == (0,)
    assert eng.score_sp(T, z, z).shape == (0, 10)
    assert eng.score_sp(T, torch.tensor([1], device=DEV), torch.tensor([0], device=DEV), z).shape == (1, 0)


# ---- rank counts ---------------------------------------------------------------------------
def test_rank_core_golden(eng):
    g = np.load(os.path.join(GOLDEN, "rankcore.npz"))
    atol, rtol = float(g["atol"]), float(g["rtol"])
    rank, ties = eng.rank_counts(_t(g["scores"]), _t(g["true"]), atol=atol, rtol=rtol)
    _eq("rank", _np(rank), g["rank"])
    _eq("ties", _np(ties), g["ties"])
    lab = g["labels"]
    n, c = g["scores"].shape

    def csr(block):
        rp, col = [0], []
        for i in range(n):
            col.extend(np.nonzero(np.isinf(block[i]))[0].tolist())
            rp.append(len(col))
        return np.array(rp, dtype=np.int64), np.array(col, dtype=np.int64)

    rp, col = csr(lab[:, :c])
    r, t = eng.rank_counts(_t(g["scores"]), _t(g["true"]), _t(rp), _t(col), atol=atol, rtol=rtol)
    _eq("filt_o_rank", _np(r), g["filt_o_rank"])
    _eq("filt_o_ties", _np(t), g["filt_o_ties"])
    rp, col = csr(lab[:, c:])
    r, t = eng.rank_counts(_t(g["scores_po"]), _t(g["true_po"]), _t(rp), _t(col), atol=atol, rtol=rtol)
    _eq("filt_s_rank", _np(r), g["filt_s_rank"])
    _eq("filt_s_ties", _np(t), g["filt_s_ties"])


@pytest.mark.parametrize("n,c,lds_pad", [(1, 1, 0), (7, 1000, 3), (33, 14541, 0), (5, 70001, 1)])
def test_rank_counts_random_vs_oracle_and_torch(eng, n, c, lds_pad):
    rng = np.random.default_rng(n * 1000 + c)
    buf = (rng.standard_normal((n, c + lds_pad)) * 3).astype(np.float32)
    sc = buf[:, :c]
    tcol = rng.integers(0, c, n)
    true = sc[np.arange(n), tcol].copy()
    # ties, near-ties, NaN and infinities
    for i in range(n):
        k = rng.integers(0, c, 5)
        sc[i, k] = true[i] + np.float32(rng.choice([0, 1e-6, -1e-6, 2e-4, 1e-5]))
    sc[0, 0] = np.nan
    if c > 3:
        sc[n - 1, 1] = np.inf
        sc[n - 1, 2] = -np.inf
    # CSR filter labels (global ids with an offset), unique per row, may contain the positive
    off = 17
    rp, col = [0], []
    for i in range(n):
        k = np.unique(np.concatenate([rng.integers(0, c, min(c, 40)), [tcol[i]]])) + off
        col.extend(k.tolist())
        rp.append(len(col))
    rp, col = np.array(rp, np.int64), np.array(col, np.int64)
    tsc = _t(buf)[:, :c]
    for use_filter in (False, True):
        kw = dict(lbl_rowptr=rp, lbl_col=col, col_offset=off, true_col=tcol + off) if use_filter else {}
        want_r, want_t = ko.rank_counts(sc, true, **kw)
        gkw = {k: (_t(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
        r, t = eng.rank_counts(tsc, _t(true), **gkw)
        _eq(f"rank filt={use_filter}", _np(r), want_r)
        _eq(f"ties filt={use_filter}", _np(t), want_t)
    # the reference's own op sequence on the GPU (eval_entity_ranking.py:583-595)
    x = tsc.clone()
    x[torch.isnan(x)] = float("-inf")
    tt = _t(true).view(-1, 1)
    close = torch.isclose(x, tt, rtol=1e-4, atol=1e-5)
    r, t = eng.rank_counts(tsc, _t(true))
    _eq("torch rank", _np(r), _np(((x > tt) & ~close).sum(1)))
    _eq("torch ties", _np(t), _np(close.sum(1)))


@pytest.mark.parametrize("n,c", [(1, 1), (7, 1000), (33, 14541), (5, 70001)])
def test_filter_lookup_multi_ranking_and_histogram_vs_oracle(eng, n, c):
    """The sync-free evaluation entries against the oracle: kge_filter_lookup on a device-resident
    sorted index (known and unknown keys), kge_rank_counts_multi (raw + two filter sets from one
    scan, accumulated over two column chunks) against one oracle rank_counts per ranking, and
    kge_rank_hist for the three tie policies against numpy.  Integer work: exact."""
    rng = np.random.default_rng(n * 7 + c)
    sc = (rng.standard_normal((n, c)) * 3).astype(np.float32)
    tcol = rng.integers(0, c, n)
    true = sc[np.arange(n), tcol].copy()
    for i in range(n):
        sc[i, rng.integers(0, c, 5)] = true[i] + np.float32(rng.choice([0, 1e-6, -1e-6, 2e-4, 1e-5]))
    sc[0, 0] = np.nan
    # an index over keys a * mult + b with a few values each; queries: known keys + one unknown
    mult = 50
    a, b = rng.integers(0, 40, n), rng.integers(0, mult, n)
    a[-1], b[-1] = 45, 3                     # never inserted
    qkeys = a * mult + b
    filters_np = []
    for f in range(2):
        keys = np.unique(np.concatenate([qkeys[:-1], rng.integers(0, 40 * mult, 30)]))
        starts, vals = [0], []
        for k in keys:
            v = np.unique(rng.integers(0, c, min(c, 10 + 20 * f)))
            hit = np.nonzero(qkeys == k)[0]
            if len(hit) and f == 0:          # the positive is among the known answers
                v = np.unique(np.append(v, tcol[hit[0]]))
            vals.extend(v.tolist())
            starts.append(len(vals))
        filters_np.append((keys.astype(np.int64), np.array(starts, np.int64), np.array(vals, np.int64)))
    filters, csrs = [], []
    for keys, starts, vals in filters_np:
        beg, end = torch.zeros(n, dtype=torch.int64, device=DEV), torch.zeros(n, dtype=torch.int64, device=DEV)
        eng.filter_lookup(_t(keys), _t(starts), _t(a), _t(b), mult, beg, end)
        pos = np.searchsorted(keys, qkeys)
        hit = (pos < len(keys)) & (keys[np.minimum(pos, len(keys) - 1)] == qkeys)
        wb = np.where(hit, starts[np.minimum(pos, len(keys) - 1)], 0)
        we = np.where(hit, starts[np.minimum(pos, len(keys) - 1) + 1], 0)
        _eq("begin", _np(beg), wb)
        _eq("end", _np(end), we)
        assert wb[-1] == 0 and we[-1] == 0
        filters.append((beg, end, _t(vals)))
        # the same lookup through the multi entry (here: twice in one launch, with int32 / strided index views)
        mb = torch.full((2, 2, n), -7, dtype=torch.int64, device=DEV)
        a2 = torch.stack([_t(a), _t(b)], 1).int()
        eng.filter_lookup_multi([(_t(keys), _t(starts), _t(a), _t(b), mult, mb[0, 0], mb[0, 1]),
                                 (_t(keys), _t(starts), a2[:, 0], a2[:, 1], mult, mb[1, 0], mb[1, 1])])
        for q in range(2):
            _eq("begin (multi)", _np(mb[q, 0]), wb)
            _eq("end (multi)", _np(mb[q, 1]), we)
        rp = np.concatenate([[0], np.cumsum(we - wb)])
        col = np.concatenate([vals[x:y] for x, y in zip(wb, we)] + [np.zeros(0, np.int64)])
        csrs.append((rp.astype(np.int64), col.astype(np.int64)))
    rank = torch.zeros(3, n, dtype=torch.int64, device=DEV)
    ties = torch.zeros(3, n, dtype=torch.int64, device=DEV)
    tsc, half = _t(sc), max(1, c // 2)
    for lo, hi in ((0, half), (half, c)):
        if hi > lo:
            eng.rank_counts_multi(tsc[:, lo:hi], _t(true), filters, lo, _t(tcol), 1e-5, 1e-4, rank, ties)
    want = [ko.rank_counts(sc, true)] + [ko.rank_counts(sc, true, lbl_rowptr=rp, lbl_col=col, col_offset=0,
                                                        true_col=tcol) for rp, col in csrs]
    for k in range(3):
        _eq(f"rank[{k}]", _np(rank[k]), want[k][0])
        _eq(f"ties[{k}]", _np(ties[k]), want[k][1])
    for policy in ("rounded_mean_rank", "best_rank", "worst_rank"):
        hist = torch.zeros(3, c, device=DEV)
        out = torch.empty(3, n, dtype=torch.int64, device=DEV)
        eng.rank_hist(rank, ties, policy, hist, out)
        r, t = _np(rank), _np(ties)
        wr = {"rounded_mean_rank": r + t // 2, "best_rank": r, "worst_rank": r + t - 1}[policy]
        _eq(policy, _np(out), wr)
        wh = np.zeros((3, c), np.float32)
        for k in range(3):
            np.add.at(wh[k], wr[k][(wr[k] >= 0) & (wr[k] < c)], 1.0)
        _eq(policy + " hist", _np(hist), wh)


def test_duplicates_zero_rows_and_non_finite_values(eng):
    """Collisions and degenerate values: repeated query indices, repeated ids in the listed
    subset, an all-zero entity row, infinities and NaNs in the tables.  f32-arithmetic paths
    bit for bit against the oracle (NaN == NaN); the bf16 matrix-core kernel: duplicates give
    identical columns / rows, finite entries within tolerance, non-finite entries where the
    oracle has them."""
    rng = np.random.default_rng(21)
    E, R, d, n = 300, 5, 256, 70
    ent = rng.standard_normal((E, d)).astype(np.float32)
    rel = rng.standard_normal((R, d)).astype(np.float32)
    ent[7] = 0.0
    ent[11, 3] = np.inf
    ent[13, 200] = np.nan
    rel[2, 5] = -np.inf
    s = rng.integers(0, E, n); s[:6] = [7, 11, 13, 7, 11, 13]; s[10:20] = s[20:30]  # repeated queries
    p = rng.integers(0, R, n); p[:6] = [0, 1, 2, 2, 0, 1]; p[10:20] = p[20:30]
    sub = np.concatenate([rng.integers(0, E, 150), [7, 11, 13, 7, 7], rng.integers(0, E, 40)])  # repeated targets
    for model in ("complex", "distmult", "transe", "rotate"):
        rl = rel[:, : d // 2] if model == "rotate" else rel
        T = _gpu_tables(eng, model, ent, rl, 1.0, flags=eng.FLAG_EXACT)
        O = _oracle_tables(model, ent, rl, 1.0)
        _eq(f"{model} sp_sub", _np(eng.score_sp(T, _t(s), _t(p), _t(sub))), ko.score_sp(O, s, p, sub))
        _eq(f"{model} spo", _np(eng.score_spo(T, _t(s), _t(p), _t(s[::-1].copy()))), ko.score_spo(O, s, p, s[::-1].copy()))
    for model in ("complex", "distmult"):
        T = _gpu_tables(eng, model, ent, rel, 1.0, bf16=True)
        O = _oracle_tables(model, ent, rel, 1.0, bf16=True)
        got, want = _np(eng.score_sp(T, _t(s), _t(p), _t(sub))), ko.score_sp(O, s, p, sub)
        fin = np.isfinite(want)
        assert np.array_equal(np.isnan(got), np.isnan(want)), model
        assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]), model  # same infinities
        g2, w2 = got.copy(), want.copy()
        g2[~fin] = 0.0
        w2[~fin] = 0.0
        _close(f"{model} bf16 finite entries", g2, w2)
        _eq(f"{model} repeated queries -> identical rows", got[10:20], got[20:30])
        first7 = int(np.nonzero(sub == 7)[0][0])
        for j in np.nonzero(sub == 7)[0][1:]:
            _eq(f"{model} repeated target -> identical columns", got[:, j], got[:, first7])


def test_short_square_root_is_correctly_rounded_for_every_float(eng):
    """common.hpp sqrt_rn_fast against the compiler's IEEE sequence over ALL 2^32 bit patterns (zeros, denormals,
    both edges of the short form's range, inf, NaNs, negatives), on the device: not one differs."""
    import ctypes
    from kge_amd import _lib
    fn = _lib.lib().kge_debug_sqrt_check
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    bad = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    first = torch.zeros(16, dtype=torch.int32, device="cuda:0")
    for lo in (0, 1 << 31):
        assert fn(lo, 1 << 31, bad.data_ptr(), first.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert int(bad) == 0, [hex(int(x) & 0xffffffff) for x in first.tolist()]


def test_rotate_square_roots_at_the_edges_of_the_short_form(eng):
    """RotatE takes one correctly rounded square root per complex coordinate.  The kernels use a short form
    (common.hpp sqrt_rn_core: v_rsq_f32 + a Newton step + Markstein's correction, checked exhaustively against the IEEE
    sequence over all 2^32 bit patterns: tools/ubench/sqrt_exhaustive.hip) for squares in [2^-96, 2^126] and the IEEE
    sequence outside -- decided per 4 x 4 micro-tile in the pair kernel, per root in the row kernels.  Here the
    squares sit on both sides of both edges, mixed inside micro-tiles: exact zeros (a query scored against itself
    under a zero rotation), 1e-52 ... 1e-28, 1e37 ... 3e38, next to ordinary values -- bit for bit against the oracle
    (glibc sqrtf)."""
    rng = np.random.default_rng(77)
    E, R, d, n = 260, 4, 128, 48
    ent = rng.standard_normal((E, d)).astype(np.float32)
    scale = np.ones((E, 1), np.float32)
    scale[0:40] = 1e-26      # squares ~1e-52: far below 2^-96
    scale[40:80] = 3e-15     # squares ~1e-29: around 2^-96 = 1.26e-29
    scale[80:120] = 6e18     # squares ~4e37 ... 3e38: around 2^126 = 8.5e37, some sums overflow to inf
    scale[120:140] = 2e18
    ent *= scale
    rel = rng.uniform(-np.pi, np.pi, (R, d // 2)).astype(np.float32)
    rel[0] = 0.0             # zero rotation: score_sp(s, 0) against entity s has |q - t| = 0 in every coordinate
    s = rng.integers(0, E, n); s[:8] = [0, 41, 81, 121, 200, 5, 45, 85]
    p = rng.integers(0, R, n); p[:8] = 0
    o = rng.integers(0, E, n); o[:8] = s[:8]
    neg = rng.integers(0, E, (n, 7)); neg[:, 0] = s
    T = _gpu_tables(eng, "rotate", ent, rel, 1.0)
    O = _oracle_tables("rotate", ent, rel, 1.0)
    with np.errstate(over="ignore", invalid="ignore"):
        _eq("sp_all", _np(eng.score_sp(T, _t(s), _t(p))), ko.score_sp(O, s, p))
        _eq("po_all", _np(eng.score_po(T, _t(p), _t(o))), ko.score_po(O, p, o))
        _eq("spo", _np(eng.score_spo(T, _t(s), _t(p), _t(o))), ko.score_spo(O, s, p, o))
        _eq("neg_o", _np(eng.score_neg(T, _t(s), _t(p), _t(o), 2, _t(neg))), ko.score_neg(O, s, p, o, 2, neg))
    assert float(_np(eng.score_spo(T, _t(s[:1]), _t(p[:1]), _t(s[:1])))[0]) == 0.0


@pytest.mark.parametrize("model,dt", [("complex", torch.bfloat16), ("rotate", torch.float32)])
def test_embed_rows(eng, model, dt):
    """kge_embed (LookupEmbedder.embed, both tables in one launch) == tensor indexing; strided
    outputs, int32 / int64 / strided index views, empty sides."""
    g = torch.Generator().manual_seed(9)
    E, R, d = 500, 7, 128
    dr = d // 2 if model == "rotate" else d
    ent = torch.randn(E, d, generator=g).to(dt).to(DEV)
    rel = torch.randn(R, dr, generator=g).to(dt).to(DEV)
    T = eng.Tables(model, ent, rel)
    tri = torch.stack([torch.randint(E, (77,), generator=g), torch.randint(R, (77,), generator=g),
                       torch.randint(E, (77,), generator=g)], 1).to(DEV)
    e, r = eng.embed(T, tri[:, 0], tri[:, 1].int())
    assert torch.equal(e, ent[tri[:, 0]]) and torch.equal(r, rel[tri[:, 1]])
    wide = torch.zeros(77, 2 * d, dtype=dt, device=DEV)
    eng.embed(T, tri[:, 2], None, wide[:, d:], None)
    assert torch.equal(wide[:, d:], ent[tri[:, 2]]) and not wide[:, :d].any()
    e2, r2 = eng.embed(T, None, tri[:5, 1])
    assert e2 is None and torch.equal(r2, rel[tri[:5, 1]])


def test_bf16_handoff_stress_under_load(eng):
    """The cooperative query build's hand-off (sc1 stores -> flag -> sc1 loads) checked on every
    word, 400 launches with fresh queries, while a side stream keeps the memory system busy with
    copies (uneven load: the published fragments compete with 256 MB transfers), alternating
    one- and two-sided launches on the same scratch buffer."""
    g = torch.Generator().manual_seed(77)
    E, R, d, n = 9000, 11, 512, 384
    ent = torch.randn(E, d, generator=g).bfloat16().to(DEV)
    rel = torch.randn(R, d, generator=g).bfloat16().to(DEV)
    T = eng.Tables("complex", ent, rel)
    Tn = eng.Tables("complex", ent, rel, use_workspace=False)
    side = torch.cuda.Stream(DEV)
    big_a = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
    big_b = torch.empty_like(big_a)
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    S = torch.randint(E, (400, n), generator=g).to(DEV)
    P = torch.randint(R, (400, n), generator=g).to(DEV)
    for it in range(400):
        if it % 4 == 0:
            with torch.cuda.stream(side):
                big_b.copy_(big_a, non_blocking=True)
        s, p = S[it], P[it]
        if it % 3 == 2:
            got = eng.score_sp_po(T, s, p, s)
            ref = torch.cat((eng.score_sp(Tn, s, p), eng.score_po(Tn, p, s)), 1)
        else:
            got, ref = eng.score_sp(T, s, p), eng.score_sp(Tn, s, p)
        bad += (got != ref).sum()
    torch.cuda.synchronize()
    assert int(bad.item()) == 0


def test_bf16_random_shapes_against_the_single_role_kernel(eng):
    """40 random shapes (rows 1..1500, targets 1..6000, d in {256, 512}, all / listed targets,
    one- and two-sided): the loader/consumer kernel with the cooperative build must give the bits
    of the no-workspace kernel (same arithmetic, independent partitioning logic), and a sampled
    row must match the oracle within the bf16 tolerance."""
    rng = np.random.default_rng(123)
    for it in range(40):
        d = int(rng.choice([256, 512]))
        E = int(rng.integers(1, 6001))
        n = int(rng.integers(1, 1501)) if it % 5 else int(rng.integers(1, 40))
        R = 5
        model = "complex" if it % 2 else "distmult"
        ent = rng.standard_normal((E, d)).astype(np.float32)
        rel = rng.standard_normal((R, d)).astype(np.float32)
        T = _gpu_tables(eng, model, ent, rel, 1.0, bf16=True)
        Tn = _gpu_tables(eng, model, ent, rel, 1.0, bf16=True)
        Tn.use_workspace = False
        s, p, o = rng.integers(0, E, n), rng.integers(0, R, n), rng.integers(0, E, n)
        sub = None if it % 3 else rng.integers(0, E, int(rng.integers(1, E + 1)))
        ts, tp, to = _t(s), _t(p), _t(o)
        tsub = None if sub is None else _t(sub)
        tag = f"it={it} {model} d={d} n={n} E={E} sub={None if sub is None else len(sub)}"
        got = _np(eng.score_sp(T, ts, tp, tsub))
        _eq("sp " + tag, got, _np(eng.score_sp(Tn, ts, tp, tsub)))
        both = _np(eng.score_sp_po(T, ts, tp, to, tsub))
        m = got.shape[1]
        _eq("two-sided sp " + tag, both[:, :m], got)
        _eq("two-sided po " + tag, both[:, m:], _np(eng.score_po(Tn, tp, to, tsub)))
        O = _oracle_tables(model, ent, rel, 1.0, bf16=True)
        i = int(rng.integers(0, n))
        _close("oracle row " + tag, got[i:i + 1], ko.score_sp(O, s[i:i + 1], p[i:i + 1], sub))


@pytest.mark.parametrize("d,n,E", [(512, 512, 14541), (256, 130, 700), (128, 300, 5000)])
def test_bf16_own_build_fallback_is_bit_identical(eng, monkeypatch, d, n, E, kge_switch):
    """A consumer workgroup of the loader/consumer kernel (d = 128: a wave of the single-role kernel, which used to
    trap here) whose builders do not show up within a bounded wait (their CUs busy with another kernel) builds
    its own query fragments and marks the workspace degraded; KGE_V4_OWN_BUILD=1 forces that path.  The scores
    must be the bits of the cooperative build, one- and two-sided, the two modes must alternate on one
    scratch buffer, and a workspace whose "degraded" word is set must keep giving the same bits."""
    from kge_amd import engine as engmod
    rng = np.random.default_rng(d + n)
    R = 9
    ent = rng.standard_normal((E, d)).astype(np.float32)
    rel = rng.standard_normal((R, d)).astype(np.float32)
    for model in ("complex", "distmult"):
        T = _gpu_tables(eng, model, ent, rel, 1.0, bf16=True)
        s, p, o = (_t(rng.integers(0, hi, n)) for hi in (E, R, E))

        def calls():
            return [_np(eng.score_sp(T, s, p)), _np(eng.score_po(T, p, o)), _np(eng.score_sp_po(T, s, p, o))]
        coop = calls()
        for rep in range(2):
            kge_switch.set("V4_OWN_BUILD", "1")
            own = calls()
            kge_switch.set("V4_OWN_BUILD", "0")
            again = calls()
            for k, (a, b, c) in enumerate(zip(coop, own, again)):
                _eq(f"{model} own build, call {k}, rep {rep}", b, a)
                _eq(f"{model} cooperative build after own build, call {k}, rep {rep}", c, a)
    # a degraded workspace: poison every scratch buffer of this process (flag lines and the degraded
    # word become non-zero garbage), score again, then restore zeros
    bufs = [b for k, b in engmod._WORKSPACES.items() if len(k) == 2]
    for b in bufs:
        b.fill_(0xA5)
    degraded = calls()
    for b in bufs:
        b.zero_()
    for k, (a, b) in enumerate(zip(coop, degraded)):
        _eq(f"degraded workspace, call {k}", b, a)


def test_workspace_control_block_survives_calls_of_other_sizes(eng):
    """One scratch buffer per stream serves calls with different row counts.  The control block (flag lines +
    the "degraded" word) sits at the START of the workspace, at an offset that does not depend on n: when it
    lay behind the query fragments, the fragments of a larger call overwrote the degraded word of a smaller
    one, and every later small call took the own-build path (same bits, 17 -> 24 us at C2).  After calls of
    several sizes, one- and two-sided, the degraded word is still zero and no flag line is left set."""
    from kge_amd import engine as engmod
    rng = np.random.default_rng(77)
    E, R, d = 3000, 7, 512
    ent = rng.standard_normal((E, d)).astype(np.float32)
    rel = rng.standard_normal((R, d)).astype(np.float32)
    T = _gpu_tables(eng, "complex", ent, rel, 1.0, bf16=True)
    ref = {}
    for rep in range(2):
        for n in (2048, 130, 512, 1024, 64):
            s, p, o = (_t(np.random.default_rng(n).integers(0, hi, n)) for hi in (E, R, E))
            got = (_np(eng.score_sp(T, s, p)), _np(eng.score_sp_po(T, s, p, o)))
            if n in ref:
                _eq(f"n={n}, repeat", got[0], ref[n][0])
                _eq(f"n={n}, repeat (two-sided)", got[1], ref[n][1])
            ref[n] = got
    torch.cuda.synchronize()
    bufs = [b for k, b in engmod._WORKSPACES.items() if len(k) == 2]
    assert bufs
    for b in bufs:
        ctrl = b[:512 * 64 + 8].cpu().numpy()
        assert not ctrl[512 * 64:].any(), "the degraded word was set (or overwritten)"
        assert not ctrl[:512 * 64].any(), "a flag line was left behind"


def test_degraded_workspace_recovers(eng):
    """The "degraded" word of the cooperative build is a countdown of launches: a workspace marked degraded (a
    hand-off timed out once) takes the own-build path for that many launches -- same bits -- and then goes back
    to the cooperative build on its own."""
    from kge_amd import engine as engmod
    rng = np.random.default_rng(78)
    E, R, d, n = 2000, 7, 512, 300
    ent = rng.standard_normal((E, d)).astype(np.float32)
    rel = rng.standard_normal((R, d)).astype(np.float32)
    T = _gpu_tables(eng, "complex", ent, rel, 1.0, bf16=True)
    s, p = _t(rng.integers(0, E, n)), _t(rng.integers(0, R, n))
    ref = _np(eng.score_sp(T, s, p))
    torch.cuda.synchronize()
    # the scratch buffer of THIS stream (other tests leave buffers of their capture streams behind)
    dev = torch.device(DEV)
    bufs = [engmod._WORKSPACES[(dev.index, engmod._stream_handle(dev))]]
    word = slice(512 * 64, 512 * 64 + 8)
    for b in bufs:
        b[word] = torch.tensor([3, 0, 0, 0, 0, 0, 0, 0], dtype=torch.uint8, device=b.device)  # little-endian 3
    seen = []
    for k in range(5):
        _eq(f"launch {k} on a degraded workspace", _np(eng.score_sp(T, s, p)), ref)
        torch.cuda.synchronize()
        seen.append(int(bufs[0][word].cpu().numpy().view(np.uint64)[0]))
    assert seen == [2, 1, 0, 0, 0], seen


def test_what_the_cooperative_kernel_declines_runs_on_the_single_role_kernel(eng, kge_switch):
    """Route 4 of api.hip's bf16_store_dispatch: calls the loader/consumer kernel (v4) does not take -- no scratch
    buffer, more than 32 row groups -- run on pairs_bf16_v3_kernel with the same bits.  (Until round 6 a fifth kernel
    generation, the workgroup-local-build kernel "v5", sat between the two; it is gone.)  Ragged shapes, all / listed
    targets, one- and two-sided, dense rows, d = 256 and 512."""
    rng = np.random.default_rng(55)
    for it in range(10):
        d = int(rng.choice([256, 512]))
        E = int(rng.integers(1, 5000))
        n = int(rng.integers(1, 1000)) if it % 4 else int(rng.integers(1, 70))
        R = 5
        model = "complex" if it % 2 else "distmult"
        ent = rng.standard_normal((E, d)).astype(np.float32)
        rel = rng.standard_normal((R, d)).astype(np.float32)
        T = _gpu_tables(eng, model, ent, rel, 1.0, bf16=True)
        Tn = _gpu_tables(eng, model, ent, rel, 1.0, bf16=True)
        Tn.use_workspace = False
        s, p, o = (_t(rng.integers(0, hi, n)) for hi in (E, R, E))
        sub = None if it % 3 else _t(rng.integers(0, E, int(rng.integers(1, E + 1))))
        want = (_np(eng.score_sp(T, s, p, sub)), _np(eng.score_po(T, p, o, sub)), _np(eng.score_sp_po(T, s, p, o, sub)))
        got = (_np(eng.score_sp(Tn, s, p, sub)), _np(eng.score_po(Tn, p, o, sub)), _np(eng.score_sp_po(Tn, s, p, o, sub)))
        dense = _np(eng.score_emb(model, T.ent[s.long()], T.rel[p.long()], T.ent if sub is None else T.ent[sub.long()], "sp_"))
        tag = f"it={it} {model} d={d} n={n} E={E} sub={None if sub is None else int(sub.numel())}"
        for k, (a, b) in enumerate(zip(got, want)):
            _eq(f"no scratch buffer, call {k}, {tag}", a, b)
        _eq(f"dense rows, {tag}", dense, want[0])
    kge_switch.set("ONE_CALL_V8", "0")  # (n >= 1024 would leave as batches of the persistent kernel otherwise)
    E, d, n = 3000, 512, 4500
    ent = rng.standard_normal((E, d)).astype(np.float32)
    rel = rng.standard_normal((R, d)).astype(np.float32)
    T = _gpu_tables(eng, "complex", ent, rel, 1.0, bf16=True)
    Tn = _gpu_tables(eng, "complex", ent, rel, 1.0, bf16=True)
    Tn.use_workspace = False
    s, p = _t(rng.integers(0, E, n)), _t(rng.integers(0, R, n))
    big = _np(eng.score_sp(T, s, p))            # 36 row groups of 128: beyond the cooperative kernel
    kge_switch.unset("ONE_CALL_V8")
    _eq("n = 4500: the single-role kernel == the persistent kernel's batches", _np(eng.score_sp(T, s, p)), big)
    _eq("n = 4500 vs two halves", big, np.concatenate([_np(eng.score_sp(T, s[:2250], p[:2250])),
                                                       _np(eng.score_sp(T, s[2250:], p[2250:]))]))
    _eq("no scratch buffer", _np(eng.score_sp(Tn, s[:700], p[:700])), big[:700])
    O = _oracle_tables("complex", ent, rel, 1.0, bf16=True)
    rows = rng.integers(0, n, 8)
    _close("vs oracle", big[rows], ko.score_sp(O, _np(s)[rows], _np(p)[rows]))


def test_torch_extension_path_equals_the_ctypes_path(monkeypatch):
    """The index-level scoring calls through kge_amd._C (the default binding) and through ctypes (KGE_AMD_BINDING=ctypes)
    reach the same C entry points: the same bits, int32 / strided indices, listed targets, both table dtypes."""
    from kge_amd import engine
    g = torch.Generator().manual_seed(3)
    E, R, d, n = 1500, 7, 256, 77
    ent, rel = torch.randn(E, d, generator=g) * 0.3, torch.randn(R, d, generator=g) * 0.3
    tri = torch.stack([torch.randint(hi, (n,), generator=g) for hi in (E, R, E)], 1).to(DEV)
    sub = torch.randperm(E, generator=g)[:333].to(torch.int32).to(DEV)
    got = {}
    for binding in ("ext", "ctypes"):
        monkeypatch.setenv("KGE_AMD_BINDING", binding)
        engine._EXT = None
        assert bool(engine._ext()) == (binding == "ext")
        res = []
        for dt in (torch.float32, torch.bfloat16):
            for model in ("complex", "transe"):
                T = engine.Tables(model, ent.to(dt).to(DEV), rel.to(dt).to(DEV))
                s, p, o = tri[:, 0], tri[:, 1].to(torch.int32), tri[:, 2]
                res += [engine.score_spo(T, s, p, o), engine.score_sp(T, s, p), engine.score_po(T, p, o, sub),
                        engine.score_sp_po(T, s, p, o), engine.score_sp_po(T, s, p, o, sub)]
        got[binding] = res
    engine._EXT = None
    for a, b in zip(got["ext"], got["ctypes"]):
        assert a.shape == b.shape and torch.equal(a, b)
