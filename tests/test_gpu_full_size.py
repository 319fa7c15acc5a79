"""GPU tests at the FULL sizes of BASELINE.json's configs (C2, C3, C5-shard).

The oracle cannot restate a 512 x 14,541 x 512 contraction (let alone the C5 shard) for
every element in seconds, so these tests use what the domain offers:

  * sampled rows against the oracle (bit-exact for f32 arithmetic, reference tolerance for
    the bf16 MFMA kernel);
  * consistency between entry points on ALL elements: the listed-subset call equals the
    columns of the all-entities call bit for bit; the two-sided score_sp_po launch equals
    the two one-sided calls; kernels selected by flags agree (bit-exact where the K order
    is the same, tolerance otherwise); score_spo equals the matching element of score_sp
    (the reference's own test, tests/test_model.py:29-71);
  * exact linearity: ComplEx / DistMult scores are linear in the relation row, and scaling
    by a power of two commutes with every rounding -> scores of (rel * 2) == 2 * scores,
    bit for bit;
  * negatives: score_neg equals score_spo on the expanded triples, bit for bit;
  * rank counts are additive over column chunks (kge/job/eval_entity_ranking.py:310-313).
"""
import numpy as np
import pytest
import torch

import oracle as ko
from test_gpu_parity import DEV, _close, _eq, _np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from kge_amd import engine
    return engine


def _tables(eng, model, E, R, d, dr=None, dtype=torch.float32, seed=0, flags=0):
    g = torch.Generator().manual_seed(seed)
    ent = torch.empty(E, d).normal_(0, 0.1, generator=g)
    if model == "rotate":
        rel = torch.empty(R, dr or d // 2).uniform_(-3.14159, 3.14159, generator=g)
    else:
        rel = torch.empty(R, dr or d).normal_(0, 0.1, generator=g)
    ent, rel = ent.to(dtype), rel.to(dtype)
    return ent, rel, eng.Tables(model, ent.to(DEV), rel.to(DEV), 1.0, flags)


def _oracle(model, ent, rel):
    """oracle tables on the values the GPU sees (bf16 tables are widened exactly)"""
    if ent.dtype == torch.bfloat16:
        to16 = lambda x: x.view(torch.int16).numpy().view(np.uint16)  # noqa: E731
        return ko.Tables(model, to16(ent), to16(rel), 1.0)
    return ko.Tables(model, ent.numpy(), rel.numpy(), 1.0)


def _q_f32(model, ent, rel, a, p, direction):
    """the query vector of row i in f32 numpy (ComplEx / DistMult), NOT rounded to bf16"""
    A, Rr = ent[a].float().numpy(), rel[p].float().numpy()
    if model == "distmult":
        return A * Rr
    h = A.shape[1] // 2
    are, aim, rre, rim = A[:, :h], A[:, h:], Rr[:, :h], Rr[:, h:]
    if direction == "sp":
        return np.concatenate([are * rre - aim * rim, aim * rre + are * rim], 1)
    return np.concatenate([rre * are + rim * aim, rre * aim - rim * are], 1)


def _assert_within_bf16_rounding(name, got, want, qnorm, tnorm):
    """|got - want| <= 2^-8 * |q| * |t| (+ f32 slack): the bf16 kernels round the query vector to
    bf16 (relative 2^-9 per element), the f32-arithmetic paths do not."""
    bound = 2.0 ** -8 * np.asarray(qnorm, np.float64) * np.asarray(tnorm, np.float64) + 1e-6
    err = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
    assert (err <= bound).all(), (name, float((err / bound).max()))


def _queries(E, R, n, seed=1):
    g = torch.Generator().manual_seed(seed)
    return (torch.randint(E, (n,), generator=g).to(DEV), torch.randint(R, (n,), generator=g).to(DEV),
            torch.randint(E, (n,), generator=g).to(DEV))


@pytest.mark.parametrize("model", ["complex", "distmult"])
def test_c2_full_size_bf16(eng, model):
    """C2: FB15k-237 shape, d=512, n=512, bf16 tables -- the benchmark workload itself."""
    E, R, d, n = 14541, 237, 512, 512
    ent, rel, T = _tables(eng, model, E, R, d, dtype=torch.bfloat16)
    s, p, o = _queries(E, R, n)
    sp, po = eng.score_sp(T, s, p), eng.score_po(T, p, o)
    # (1) sampled rows against the oracle
    O = _oracle(model, ent, rel)
    rows = np.array([0, 1, 31, 32, 127, 128, 300, 511])
    sn, pn, on = _np(s)[rows], _np(p)[rows], _np(o)[rows]
    _close("sp rows vs oracle", _np(sp)[rows], ko.score_sp(O, sn, pn))
    _close("po rows vs oracle", _np(po)[rows], ko.score_po(O, pn, on))
    # (2) every element: the other kernels and paths
    Tn = eng.Tables(model, T.ent, T.rel, 1.0, use_workspace=False)
    _eq("no-workspace kernel == default", _np(eng.score_sp(Tn, s, p)), _np(sp))
    _eq("v3 == default", _np(eng.score_sp(T, s, p, flags=eng.FLAG_BF16_V3)), _np(sp))
    _eq("v3 == default (po)", _np(eng.score_po(T, p, o, flags=eng.FLAG_BF16_V3)), _np(po))
    _close("exact f32-chain twin ~ default", _np(eng.score_sp(T, s, p, flags=eng.FLAG_EXACT)), _np(sp))
    both = eng.score_sp_po(T, s, p, o)
    _eq("two-sided launch, sp block", _np(both[:, :E]), _np(sp))
    _eq("two-sided launch, po block", _np(both[:, E:]), _np(po))
    # (3) listed subset (ragged length, permuted) == columns of the all-entities call
    sub = torch.randperm(E, generator=torch.Generator().manual_seed(2))[: 64 * 100 + 37].to(DEV)
    _eq("subset columns", _np(eng.score_sp(T, s, p, sub)), _np(sp[:, sub]))
    _eq("subset columns (int32 ids)", _np(eng.score_po(T, p, o, sub.int())), _np(po[:, sub]))
    # (4) spo (f32 arithmetic on the bf16 table values, q not rounded) vs the matching element of
    # sp_ / _po (reference tests/test_model.py:29-71), within the rounding of q to bf16; and the
    # sampled rows against f32 arithmetic on the same table values
    spo = _np(eng.score_spo(T, s, p, o))
    ar = torch.arange(n, device=DEV)
    sc, pc, oc = s.cpu(), p.cpu(), o.cpu()
    entn = np.linalg.norm(ent.float().numpy().astype(np.float64), axis=1)
    qsp = np.linalg.norm(_q_f32(model, ent, rel, sc, pc, "sp").astype(np.float64), axis=1)
    qpo = np.linalg.norm(_q_f32(model, ent, rel, oc, pc, "po").astype(np.float64), axis=1)
    _assert_within_bf16_rounding("spo vs sp_[i, o_i]", spo, _np(sp[ar, o]), qsp, entn[oc.numpy()])
    _assert_within_bf16_rounding("spo vs _po[i, s_i]", spo, _np(po[ar, s]), qpo, entn[sc.numpy()])
    Of = ko.Tables(model, ent.float().numpy(), rel.float().numpy(), 1.0)  # f32 arithmetic, same values
    _assert_within_bf16_rounding("sp rows vs f32 arithmetic", _np(sp)[rows], ko.score_sp(Of, sn, pn),
                                 qsp[rows][:, None], entn[None, :])
    # (5) exact linearity in the relation row
    T2 = eng.Tables(model, T.ent, (T.rel.float() * 2).to(torch.bfloat16), 1.0)
    _eq("scores(2*rel) == 2*scores", _np(eng.score_sp(T2, s, p)), 2.0 * _np(sp))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("model", ["complex", "transe", "rotate"])
def test_c2_full_size_exact_paths(eng, model, dtype):
    """C2 shape through the f32-arithmetic kernels (bit-exact class): sampled rows against the
    oracle, subset == columns, MFMA chain == VALU chain."""
    E, R, d, n = 14541, 237, 512, 128
    ent, rel, T = _tables(eng, model, E, R, d, dtype=dtype, flags=eng.FLAG_EXACT)
    s, p, o = _queries(E, R, n)
    sp = eng.score_sp(T, s, p)
    O = _oracle(model, ent, rel)
    rows = np.array([0, 63, 64, 127])
    _eq("sp rows vs oracle", _np(sp)[rows], ko.score_sp(O, _np(s)[rows], _np(p)[rows]))
    sub = torch.randperm(E, generator=torch.Generator().manual_seed(3))[:4097].to(DEV)
    _eq("subset columns", _np(eng.score_sp(T, s, p, sub)), _np(sp[:, sub]))
    if model == "complex":
        _eq("f32 MFMA chain == VALU chain", _np(eng.score_sp(T, s, p, flags=eng.FLAG_EXACT | eng.FLAG_NO_MFMA)), _np(sp))


@pytest.mark.parametrize("model", ["rotate", "transe", "complex"])
def test_c3_full_size_negatives(eng, model):
    """C3: WN18RR shape, d=512, n=512, K=1000 uniform negatives per slot, f32 tables."""
    E, R, d, n, K = 40943, 11, 512, 512, 1000
    ent, rel, T = _tables(eng, model, E, R, d)
    s, p, o = _queries(E, R, n)
    neg = torch.randint(E, (n, K), generator=torch.Generator().manual_seed(4)).to(DEV)
    O = _oracle(model, ent, rel)
    rows = np.array([0, 255, 511])
    for slot in (0, 2):
        got = eng.score_neg(T, s, p, o, slot, neg)
        ss = neg.reshape(-1) if slot == 0 else s.repeat_interleave(K)
        oo = neg.reshape(-1) if slot == 2 else o.repeat_interleave(K)
        _eq(f"slot {slot}: score_neg == score_spo on the expanded triples", _np(got).reshape(-1),
            _np(eng.score_spo(T, ss, p.repeat_interleave(K), oo)))
        _eq(f"slot {slot}: rows vs oracle", _np(got)[rows],
            ko.score_neg(O, _np(s)[rows], _np(p)[rows], _np(o)[rows], slot, _np(neg)[rows]))


def test_c5_shard_full_size(eng):
    """C5: one of 8 shards of the Wikidata5M shape (574,311 entity rows, d=256, bf16), n=512:
    a 1.18 GB score slab.  Chunked calls reproduce the columns; rank counts are additive over
    chunks; sampled elements against the oracle."""
    E, R, d, n = 574311, 822, 256, 512
    ent, rel, T = _tables(eng, "complex", E, R, d, dtype=torch.bfloat16)
    s, p, o = _queries(E, R, n)
    slab = eng.score_sp(T, s, p)
    assert slab.shape == (n, E)
    bounds = [0, 100000, 100000 + 64 * 1000 + 1, 400003, E]
    rank_sum = torch.zeros(n, dtype=torch.int64, device=DEV)
    ties_sum = torch.zeros(n, dtype=torch.int64, device=DEV)
    true = slab[torch.arange(n, device=DEV), o].clone()
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        ids = torch.arange(lo, hi, device=DEV)
        part = eng.score_sp(T, s, p, ids)
        _eq(f"columns [{lo},{hi})", _np(part), _np(slab[:, lo:hi]))
        # the same chunk as a RANGE of the table (kge_index.start: the all-entities kernels, no index read)
        assert torch.equal(eng.score_sp(T, s, p, range(lo, hi)), part), f"range({lo}, {hi})"
        r, t = eng.rank_counts(part, true)
        rank_sum += r
        ties_sum += t
    r, t = eng.rank_counts(slab, true)
    _eq("rank additive over chunks", _np(rank_sum), _np(r))
    _eq("ties additive over chunks", _np(ties_sum), _np(t))
    # sampled (row, column block) against the oracle
    O = _oracle("complex", ent, rel)
    rows = np.array([0, 200, 511])
    cols = np.concatenate([np.arange(0, 64), np.arange(E - 70, E), np.array([123456, 300000])])
    _close("sampled elements vs oracle", _np(slab)[np.ix_(rows, cols)],
           ko.score_sp(O, _np(s)[rows], _np(p)[rows], cols))
    # ranks of the sampled rows against a plain recount
    x = _np(slab)[rows].astype(np.float32)
    tt = _np(true)[rows][:, None]
    close = (x == tt) | (np.abs(x - tt) <= 1e-5 + np.abs(1e-4 * tt))
    _eq("rank recount", _np(r)[rows], ((x > tt) & ~close).sum(1))
    _eq("ties recount", _np(t)[rows], close.sum(1))


def test_c5_shard_full_size_groups_in_parity_mode(eng, kge_switch):
    """C5 in parity mode (split queries) on the d = 256 persistent store kernel (round 6:
    pairs_bf16_v8_ce_kernel<128, V3_STORE, ., SPLIT>): a group of two one-sided batches of 512 against the 574,311-row
    shard -- 2 x 1.18 GB of scores from one launch -- equals one launch per batch on the single-batch split kernel bit for
    bit; sampled elements against float32 arithmetic on the bf16 values (the oracle on the widened tables)."""
    E, R, d, n, L = 574311, 822, 256, 512, 2
    ent, rel, T = _tables(eng, "complex", E, R, d, dtype=torch.bfloat16)
    fl = eng.FLAG_SPLIT_QUERY
    Ts = eng.Tables("complex", T.ent, T.rel, flags=fl)
    g = torch.Generator().manual_seed(77)
    tri = torch.stack([torch.randint(hi, (n * L,), generator=g) for hi in (E, R, E)], 1).to(DEV)
    P = eng.score_pitch(E)
    out = torch.empty(L, n, P, device=DEV)
    q = eng.build_queries_group(Ts, "sp_", tri, n, L, flags=fl)
    eng.score_queries_group(Ts, q, out[:, :, :E])
    torch.cuda.synchronize()
    kge_switch.set("V8", "0")   # one launch per batch on the single-batch kernels
    for l in range(L):
        t = tri[l * n:(l + 1) * n]
        one = eng.score_queries(Ts, eng.build_queries(Ts, "sp_", t[:, 0], t[:, 1], None, flags=fl))
        assert torch.equal(out[l, :, :E], one), f"batch {l}"
        del one
    kge_switch.unset("V8")
    Of = ko.Tables("complex", ent.float().numpy(), rel.float().numpy(), 1.0)   # the bf16 values, widened exactly
    rows = np.array([0, 300, 511])
    cols = np.concatenate([np.arange(0, 40), np.arange(E - 40, E), np.array([123456, 300000])])
    t1 = tri[n:2 * n].cpu().numpy()
    want = ko.score_sp(Of, t1[rows, 0], t1[rows, 1], cols).astype(np.float64)
    got = out[1, :, :E].cpu().numpy()[np.ix_(rows, cols)].astype(np.float64)
    big = max(1.0, float(np.abs(want).max()))
    assert np.abs(got - want).max() <= 4e-6 * big   # float32 summation noise, far inside one rounded query vector's 2^-9
