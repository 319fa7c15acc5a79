"""GPU tests of the host-side mirror of the reference API (kge_amd.model, kge_amd.eval):
they read like the reference's own tests (tests/test_model.py) and add the golden
EntityRankingJob comparison and autograd checks."""
import json
import os

import numpy as np
import pytest
import torch

import oracle as ko
import torch_port as tp
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(name, E, R, d, **kw):
    from kge_amd import model
    torch.manual_seed(0)
    return model.create(name, E, R, d, device=DEV, **kw)


@pytest.mark.parametrize("name,opts", [("complex", {}), ("distmult", {}), ("transe", {}),
                                       ("transe", {"l_norm": 2.0}), ("rotate", {}),
                                       ("rotate", {"l_norm": 2.0})])
def test_score_equality(name, opts):
    """The reference's BaseTestModel.test_score_equality (tests/test_model.py:29-71):
    score_spo(direction=o) ~ score_sp and score_spo(direction=s) ~ score_po.t()."""
    E, R = 4, 3
    m = _model(name, E, R, 32, **opts).eval()
    s = torch.arange(E).repeat_interleave(R * E).to(DEV)
    p = torch.arange(R).repeat_interleave(E).repeat(E).to(DEV)
    o = torch.arange(E).repeat(R * E).to(DEV)
    with torch.no_grad():
        spo_s = m.score_spo(s, p, o, direction="s")
        spo_o = m.score_spo(s, p, o, direction="o")
        s2 = torch.arange(E).repeat_interleave(R).to(DEV)
        p2 = torch.arange(R).repeat(E).to(DEV)
        sp = m.score_sp(s2, p2).contiguous()
        assert torch.allclose(spo_o.view(-1), sp.view(-1), atol=1e-5, rtol=1e-4)
        p3 = torch.arange(R).repeat_interleave(E).to(DEV)
        o3 = torch.arange(E).repeat(R).to(DEV)
        po = m.score_po(p3, o3).t().contiguous()
        assert torch.allclose(spo_s.view(-1), po.view(-1), atol=1e-5, rtol=1e-4)


def test_rotate_normalize_phases_keeps_scores():
    """tests/test_model.py:132-167."""
    m = _model("rotate", 20, 4, 32).eval()
    with torch.no_grad():
        m.get_p_embedder().weight.mul_(3.0)  # phases outside [-pi, pi)
        s, p, o = (torch.randint(hi, (50,), device=DEV) for hi in (20, 4, 20))
        before = m.score_spo(s, p, o)
        m.normalize_phases()
        w = m.get_p_embedder().weight
        assert (w >= -np.pi).all() and (w < np.pi + 1e-6).all()
        assert torch.allclose(before, m.score_spo(s, p, o), atol=1e-4, rtol=1e-4)


def test_state_dict_names_and_cpu_refusal():
    from kge_amd import model
    m = model.create("complex", 10, 3, 16)  # on CPU
    assert list(m.state_dict().keys()) == ["_entity_embedder._embeddings.weight",
                                           "_relation_embedder._embeddings.weight"]
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.score_sp(torch.tensor([1]), torch.tensor([0]))
    with pytest.raises(ValueError):
        model.create("rotate", 10, 3, 15)


@pytest.mark.parametrize("name", ["complex", "distmult", "transe", "rotate"])
def test_score_so_against_the_oracle(name):
    """KgeModel.score_so (kge_model.py:727-747 -> the generic s_o fallback :202-209): column j is
    score_spo with relation j -- compared with the C oracle's score_spo bit for bit (the fallback
    runs the spo kernel on repeated rows, the same chain) and with the reference's op sequence
    (oracle/torch_port.py) at the reference's own tolerance."""
    E, R, d, n = 23, 5, 16, 7
    m = _model(name, E, R, d).eval()
    ent = m.get_s_embedder().weight.detach().cpu()
    rel = m.get_p_embedder().weight.detach().cpu()
    g = torch.Generator().manual_seed(8)
    s, o = torch.randint(E, (n,), generator=g), torch.randint(E, (n,), generator=g)
    O = ko.Tables(name, ent.numpy(), rel.numpy(), 1.0)
    with torch.no_grad():
        so = m.score_so(s.to(DEV), o.to(DEV)).cpu()
        some = m.score_so(s.to(DEV), o.to(DEV), torch.tensor([3, 1], device=DEV)).cpu()
    assert so.shape == (n, R) and some.shape == (n, 2)
    for j in range(R):
        pj = np.full(n, j, dtype=np.int64)
        want = ko.score_spo(O, s.numpy(), pj, o.numpy())
        assert np.array_equal(so[:, j].numpy(), want), (name, j)
        ref = tp.score_spo(name, ent, rel, s, torch.from_numpy(pj), o)
        assert torch.allclose(so[:, j], ref, atol=1e-5, rtol=1e-4)
    assert torch.equal(some[:, 0], so[:, 3]) and torch.equal(some[:, 1], so[:, 1])


def test_unknown_combine_and_mismatched_index_lengths():
    m = _model("distmult", 12, 5, 16).eval()
    s, o = torch.tensor([1, 2], device=DEV), torch.tensor([3, 4], device=DEV)
    with torch.no_grad():
        with pytest.raises(ValueError, match='cannot handle combine="xx"'):
            m.get_scorer().score_emb(m.get_s_embedder().embed(s), m.get_p_embedder().embed(s),
                                     m.get_o_embedder().embed(o), "xx")
        with pytest.raises(ValueError, match="different lengths"):
            m.score_spo(s, torch.tensor([0], device=DEV), o)
        with pytest.raises(ValueError, match="different lengths"):
            m.score_sp(s, torch.tensor([0, 1, 2], device=DEV))


@pytest.mark.parametrize("name", ["complex", "distmult", "transe", "rotate"])
@pytest.mark.parametrize("tag,chunk", [("full", -1), ("chunk17", 17)])
def test_entity_ranking_matches_reference_golden(name, tag, chunk):
    """EntityRankingJob on the golden synthetic dataset: identical per-example ranks for the
    raw / filtered / filtered_with_test rankings, MRR and Hits within 1e-5."""
    from kge_amd import model
    from kge_amd.eval import EntityRankingEvaluator
    g = np.load(os.path.join(GOLDEN, f"eval_{name}.npz"))
    E, R = int(g["num_entities"]), int(g["num_relations"])
    m = model.create(name, E, R, g["ent"].shape[1], device=DEV).eval()
    with torch.no_grad():
        m.get_s_embedder().weight.copy_(torch.from_numpy(g["ent"]))
        m.get_p_embedder().weight.copy_(torch.from_numpy(g["rel"]))
    splits = {k: g[k] for k in ("train", "valid", "test")}
    ev = EntityRankingEvaluator(m, splits, E, R, eval_split="valid", batch_size=16, chunk_size=chunk)
    metrics, ranks = ev.run(return_ranks=True)
    for key in ("_raw", "_filt", "_filt_test"):
        gk = "" if key == "_raw" else key
        assert np.array_equal(ranks["o" + key], g[f"o_rank{gk}_{tag}"]), (name, key, "o")
        assert np.array_equal(ranks["s" + key], g[f"s_rank{gk}_{tag}"]), (name, key, "s")
    ref = json.loads(str(g[f"metrics_{tag}"]))
    for k, v in metrics.items():
        assert abs(v - ref[k]) <= 1e-5, (k, v, ref[k])


# ---- autograd --------------------------------------------------------------------------------
@pytest.mark.parametrize("name,l_norm", [("complex", 1.0), ("transe", 2.0), ("rotate", 1.0)])
def test_score_neg_blocks_is_the_composed_calls_in_one_node(name, l_norm):
    """KgeModel.score_neg_blocks (round 5): positives + both slots' negative blocks of a negative-sampling subbatch
    (train_negative_sampling.py:120-151) as ONE autograd node whose backward fills ONE pair of table gradients: the
    values are the single calls' bit for bit, the gradients of a random functional equal the composed calls' (sums in
    another order: float rounding), one slot may be absent."""
    E, R, d, n = 300, 5, 64, 41
    m = _model(name, E, R, d, l_norm=l_norm).train()
    g = torch.Generator().manual_seed(12)
    s, p, o = (torch.randint(hi, (n,), generator=g).to(DEV) for hi in (E, R, E))
    neg_s, neg_o = torch.randint(E, (n, 17), generator=g).to(DEV), torch.randint(E, (n, 9), generator=g).to(DEV)
    w_s, w_o, w_p = torch.randn(n, 17, generator=g).to(DEV), torch.randn(n, 9, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    pos, sc_s, sc_o = m.score_neg_blocks(s, p, o, neg_s, neg_o)
    assert torch.equal(pos, m.score_spo(s, p, o)) and torch.equal(sc_s, m.score_neg(s, p, o, 0, neg_s))
    assert torch.equal(sc_o, m.score_neg(s, p, o, 2, neg_o))
    m.zero_grad()
    ((pos * w_p).sum() + (sc_s * w_s).sum() + (sc_o * w_o).sum() + (pos * 2.0).sum()).backward()
    got = [q.grad.clone() for q in m.parameters()]
    m.zero_grad()
    ((m.score_spo(s, p, o) * (w_p + 2.0)).sum() + (m.score_neg(s, p, o, 0, neg_s) * w_s).sum()
     + (m.score_neg(s, p, o, 2, neg_o) * w_o).sum()).backward()
    for a, b in zip(got, [q.grad for q in m.parameters()]):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
    pos2, none_s, sc_o2 = m.score_neg_blocks(s, p, o, None, neg_o)
    assert none_s is None and torch.equal(sc_o2, sc_o)
    m.zero_grad()
    (sc_o2 * w_o).sum().backward()   # (no gradient reaches the positives)
    assert all(torch.isfinite(q.grad).all() for q in m.parameters())


@pytest.mark.parametrize("name,l_norm,d", [("complex", 1.0, 64), ("distmult", 1.0, 33), ("transe", 1.0, 130),
                                           ("transe", 2.0, 64), ("rotate", 1.0, 256), ("rotate", 2.0, 66),
                                           ("rotate", 3.0, 64)])
def test_score_neg_backward_sorted_by_entity(name, l_norm, d, monkeypatch):
    """kge_score_neg_bwd_accum_sorted (round 5): the occurrences sorted by the entity they corrupt, an entity's gradient
    row summed in registers over a run of equal ids and flushed once -- instead of one float atomic per element and
    occurrence.  Forced on (engine.NEG_BWD_SORTED = True) at shapes with many occurrences per entity, runs longer than a wave's
    chunk of 32, chunks that start and end inside a run, int32 / int64 samples, both slots: both table gradients against
    torch autograd through the reference's op sequence on the expanded triples (BatchNegativeSample.score "triple",
    kge/util/sampler.py:291-306) at the bar of the atomic kernel's test, and against the atomic kernel itself."""
    from kge_amd import engine as _engine
    E, R, n, K = 23, 3, 37, 131   # 4,847 occurrences over 23 entities: runs of ~210
    m = _model(name, E, R, d, l_norm=l_norm).train()
    ent0 = m.get_s_embedder().weight.detach().cpu().clone()
    rel0 = m.get_p_embedder().weight.detach().cpu().clone()
    g = torch.Generator().manual_seed(8)
    s, p, o = (torch.randint(hi, (n,), generator=g) for hi in (E, R, E))
    for slot, idt in ((0, torch.int32), (2, torch.int64)):
        neg = torch.randint(E, (n, K), generator=g)
        neg[:, :5] = 7                       # a hub entity: every positive draws it five times
        w = torch.randn(n, K, generator=g)
        ent, rel = ent0.clone().requires_grad_(), rel0.clone().requires_grad_()
        tr = [x.repeat_interleave(K) for x in (s, p, o)]
        tr[slot] = neg.reshape(-1)
        ref = tp.score_spo(name, ent, rel, tr[0], tr[1], tr[2], l_norm).view(n, K)
        (ref * w).sum().backward()
        grads = {}
        for mode in ("1", "0"):
            monkeypatch.setattr(_engine, "NEG_BWD_SORTED", mode == "1")
            m.zero_grad()
            got = m.score_neg(s.to(DEV), p.to(DEV), o.to(DEV), slot, neg.to(DEV).to(idt))
            (got * w.to(DEV)).sum().backward()
            grads[mode] = (m.get_s_embedder().weight.grad.cpu().clone(), m.get_p_embedder().weight.grad.cpu().clone())
        for gv, want, nm in ((grads["1"][0], ent.grad, "entity"), (grads["1"][1], rel.grad, "relation")):
            scale = max(1.0, float(want.abs().max()))
            err = float((gv - want).abs().max())
            assert err <= 2e-4 * scale, (name, l_norm, slot, nm, err, scale)
        for a, b, nm in ((grads["1"][0], grads["0"][0], "entity"), (grads["1"][1], grads["0"][1], "relation")):
            scale = max(1.0, float(b.abs().max()))
            assert float((a - b).abs().max()) <= 5e-5 * scale, (name, l_norm, slot, nm)
    monkeypatch.setattr(_engine, "NEG_BWD_SORTED", None)
    from kge_amd import engine
    assert engine._neg_bwd_sorted(512, 1000, 40943) and not engine._neg_bwd_sorted(512, 100, 14541)


@pytest.mark.parametrize("name,l_norm", [("complex", 1.0), ("distmult", 1.0), ("transe", 1.0),
                                         ("transe", 2.0), ("rotate", 1.0), ("rotate", 2.0)])
@pytest.mark.parametrize("d", [40, 33])
def test_score_neg_forward_and_backward(name, l_norm, d):
    """KgeModel.score_neg = BatchNegativeSample.score, implementation "triple" (sampler.py:291-306):
    forward bit-identical to the C oracle; gradients of a random linear functional w.r.t. both
    tables (kge_score_neg_bwd_accum) against torch autograd through the reference's op sequence on
    the expanded triples, for both corruptible slots, int32 and int64 samples, duplicates included."""
    if name in ("complex", "rotate") and d % 2:
        pytest.skip("even dimensionality only")
    E, R, n, K = 60, 4, 19, 70  # K > 64: more than one chunk of negatives per positive
    m = _model(name, E, R, d, l_norm=l_norm).train()
    ent0 = m.get_s_embedder().weight.detach().cpu().clone()
    rel0 = m.get_p_embedder().weight.detach().cpu().clone()
    g = torch.Generator().manual_seed(4)
    s, p, o = (torch.randint(hi, (n,), generator=g) for hi in (E, R, E))
    O = ko.Tables(name, ent0.numpy(), rel0.numpy(), l_norm)
    for slot, idt in ((0, torch.int64), (2, torch.int32)):
        neg = torch.randint(E, (n, K), generator=g)
        w = torch.randn(n, K, generator=g)
        ent, rel = ent0.clone().requires_grad_(), rel0.clone().requires_grad_()
        tr = [x.repeat_interleave(K) for x in (s, p, o)]
        tr[slot] = neg.reshape(-1)
        ref = tp.score_spo(name, ent, rel, tr[0], tr[1], tr[2], l_norm).view(n, K)
        (ref * w).sum().backward()
        m.zero_grad()
        got = m.score_neg(s.to(DEV), p.to(DEV), o.to(DEV), slot, neg.to(DEV).to(idt))
        assert np.array_equal(got.detach().cpu().numpy(),
                              ko.score_neg(O, s.numpy(), p.numpy(), o.numpy(), slot, neg.numpy()))
        (got * w.to(DEV)).sum().backward()
        for gv, want, nm in ((m.get_s_embedder().weight.grad.cpu(), ent.grad, "entity"),
                             (m.get_p_embedder().weight.grad.cpu(), rel.grad, "relation")):
            scale = max(1.0, float(want.abs().max()))
            err = float((gv - want).abs().max())
            assert err <= 2e-4 * scale, (name, l_norm, slot, nm, err, scale)


@pytest.mark.parametrize("name,l_norm", [("complex", 1.0), ("distmult", 1.0), ("transe", 1.0),
                                         ("transe", 2.0), ("rotate", 1.0), ("rotate", 2.0)])
def test_backward_matches_torch_autograd(name, l_norm):
    """Gradients of a random linear functional of score_sp / score_po / score_spo w.r.t. both
    tables against torch autograd through the reference's op sequence (oracle/torch_port.py,
    CPU, float64-free): tolerance level (autograd's reduction order is unspecified)."""
    E, R, d, n = 150, 5, 40, 37
    m = _model(name, E, R, d, l_norm=l_norm).train()
    ent0 = m.get_s_embedder().weight.detach().cpu().clone()
    rel0 = m.get_p_embedder().weight.detach().cpu().clone()
    g = torch.Generator().manual_seed(3)
    s, p, o = (torch.randint(hi, (n,), generator=g) for hi in (E, R, E))
    sub = torch.randperm(E, generator=g)[:45]
    w_sp = torch.randn(n, E, generator=g)
    w_po = torch.randn(n, 45, generator=g)
    w_spo = torch.randn(n, generator=g)

    # reference gradients on CPU
    ent, rel = ent0.clone().requires_grad_(), rel0.clone().requires_grad_()
    loss = ((tp.score_sp(name, ent, rel, s, p, None, l_norm) * w_sp).sum()
            + (tp.score_po(name, ent, rel, p, o, sub, l_norm) * w_po).sum()
            + (tp.score_spo(name, ent, rel, s, p, o, l_norm) * w_spo).sum())
    loss.backward()

    sd, pd, od, subd = (x.to(DEV) for x in (s, p, o, sub))
    loss2 = ((m.score_sp(sd, pd) * w_sp.to(DEV)).sum()
             + (m.score_po(pd, od, subd) * w_po.to(DEV)).sum()
             + (m.score_spo(sd, pd, od) * w_spo.to(DEV)).sum())
    loss2.backward()
    ge = m.get_s_embedder().weight.grad.cpu()
    gr = m.get_p_embedder().weight.grad.cpu()
    assert abs(loss.item() - loss2.item()) <= 1e-3 * max(1.0, abs(loss.item()))
    for got, want, nm in ((ge, ent.grad, "entity"), (gr, rel.grad, "relation")):
        scale = max(1.0, float(want.abs().max()))
        err = float((got - want).abs().max())
        assert err <= 2e-4 * scale, (name, l_norm, nm, err, scale)


def test_mixed_precision_scoring():
    """score_dtype=bfloat16 on f32 parameters (ComplEx / DistMult): the forward is the bf16
    matrix-core kernel on bf16 copies of the tables (bit for bit), the backward runs on the bf16
    copies too (gradients of the f32 model within bf16 rounding), and the copies follow the
    parameters."""
    from kge_amd import engine as eng
    from kge_amd import model as km
    E, R, d, n = 1500, 9, 256, 130
    for name in ("complex", "distmult"):
        torch.manual_seed(0)
        m32 = km.create(name, E, R, d, device=DEV)
        mmp = km.create(name, E, R, d, device=DEV, score_dtype=torch.bfloat16)
        mmp.load_state_dict(m32.state_dict())
        g = torch.Generator().manual_seed(1)
        s, p, o = (torch.randint(hi, (n,), generator=g).to(DEV) for hi in (E, R, E))
        w = torch.randn(n, E, generator=g).to(DEV)
        ent16 = mmp._entity_embedder.weight.detach().bfloat16()
        rel16 = mmp._relation_embedder.weight.detach().bfloat16()
        T16 = eng.Tables(name, ent16, rel16)
        for fn, args, ref in ((mmp.score_sp, (s, p), eng.score_sp(T16, s, p)),
                              (mmp.score_po, (p, o), eng.score_po(T16, p, o))):
            out = fn(*args)
            assert torch.equal(out, ref), name
            out32 = getattr(m32, fn.__name__)(*args)
            for mod in (m32, mmp):
                mod.zero_grad()
            (out * w).sum().backward()
            (out32 * w).sum().backward()
            for a, b in ((mmp._entity_embedder.weight.grad, m32._entity_embedder.weight.grad),
                         (mmp._relation_embedder.weight.grad, m32._relation_embedder.weight.grad)):
                # backward of the mixed-precision model runs on the bf16 matrix cores (operands
                # rounded to bf16): agreement with the f32 backward within that rounding
                torch.testing.assert_close(a, b, rtol=2e-2, atol=2e-2 * float(b.abs().max()))
                assert float((a - b).abs().mean()) <= 2e-3 * float(b.abs().max())
        # evaluation path (no autograd): two-sided launch on the copies, refreshed after an update
        with torch.no_grad():
            assert torch.equal(mmp.score_sp_po(s, p, o), eng.score_sp_po(T16, s, p, o))
            mmp._entity_embedder.weight.mul_(0.5)
            T16b = eng.Tables(name, mmp._entity_embedder.weight.detach().bfloat16(), rel16)
            assert torch.equal(mmp.score_sp_po(s, p, o), eng.score_sp_po(T16b, s, p, o))


@pytest.mark.parametrize("name", ["complex", "distmult"])
def test_bf16_backward_vs_f32_autograd(name):
    """kge_score_pairs_bwd on bf16 tables (both products on the bf16 matrix cores, gout rounded to
    bf16) against torch autograd of the reference op sequence in f32 on the same table values:
    within bf16 rounding of the operands (relative to the size of the gradient rows)."""
    from kge_amd import engine as eng
    E, R, d, n = 1100 + 3, 7, 256, 130
    g = torch.Generator().manual_seed(5)
    ent = torch.empty(E, d).normal_(0, 0.3, generator=g).bfloat16().to(DEV)
    rel = torch.empty(R, d).normal_(0, 0.3, generator=g).bfloat16().to(DEV)
    s, p, o = (torch.randint(hi, (n,), generator=g).to(DEV) for hi in (E, R, E))
    sub = torch.randperm(E, generator=g)[:257].to(DEV)
    T = eng.Tables(name, ent, rel)
    for direction, a, targets in (("sp", s, None), ("po", o, None), ("sp", s, sub)):
        m = E if targets is None else sub.numel()
        gout = torch.randn(n, m, generator=g).to(DEV)
        g_a, g_p, g_t = eng.score_pairs_bwd(T, direction, a, p, targets, gout)
        ef, rf = ent.float().requires_grad_(True), rel.float().requires_grad_(True)
        tgt = ef if targets is None else ef[targets]
        args = (ef[a], rf[p], tgt, "sp_") if direction == "sp" else (tgt, rf[p], ef[a], "_po")
        (tp.score_emb(name, *args, 1.0) * gout).sum().backward()
        want_t = ef.grad.clone()
        # reference grads of the gathered rows: subtract the query rows' own contribution is not
        # possible, so compare scatter-added tables
        ge = torch.zeros_like(want_t)
        ge.index_add_(0, a, g_a)
        if targets is None:
            ge += g_t
        else:
            ge.index_add_(0, targets, g_t)
        gr = torch.zeros_like(rf.grad)
        gr.index_add_(0, p, g_p)
        for got, want in ((ge, want_t), (gr, rf.grad)):
            scale = float(want.abs().max())
            torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2 * scale)
            # and much tighter on average
            assert float((got - want).abs().mean()) <= 2e-3 * scale


def test_sharded_table_on_the_gpu_kernels():
    """kge_amd.sharded.ShardedEntityTable with the real engine as backend (one rank, no process
    group: the exchange steps degenerate to identities): ranks / ties equal the unsharded
    engine path, for f32 (exact kernels) and bf16 (matrix-core kernel) tables, with filter labels."""
    from kge_amd import engine as eng
    from kge_amd.sharded import ShardedEntityTable
    g = torch.Generator().manual_seed(4)
    E, R, d, n = 700, 6, 256, 90
    tri = torch.stack([torch.randint(E, (n,), generator=g), torch.randint(R, (n,), generator=g),
                       torch.randint(E, (n,), generator=g)], 1).to(DEV)
    # CSR labels: a few filtered entities per row (the true entity among them)
    cnt = torch.randint(1, 6, (n,), generator=g)
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(cnt, 0)
    col_sp = torch.randint(E, (int(rowptr[-1]),), generator=g)
    col_po = torch.randint(E, (int(rowptr[-1]),), generator=g)
    col_sp[rowptr[:-1]] = tri[:, 2].cpu()
    col_po[rowptr[:-1]] = tri[:, 0].cpu()
    labels = tuple(x.to(DEV) for x in (rowptr, col_sp, rowptr, col_po))
    for dt in (torch.float32, torch.bfloat16):
        ent = torch.randn(E, d, generator=g).to(dt).to(DEV)
        rel = torch.randn(R, d, generator=g).to(dt).to(DEV)
        sh = ShardedEntityTable("distmult", ent, rel, E)
        s_rank, s_ties, o_rank, o_ties = sh.rank_batch(tri, labels)
        T = eng.Tables("distmult", ent, rel)
        s, p, o = tri[:, 0], tri[:, 1], tri[:, 2]
        both = eng.score_sp_po(T, s, p, o)
        ar = torch.arange(n, device=DEV)
        o_true, s_true = both[ar, o], both[ar, E + s]
        r_o, t_o = eng.rank_counts(both[:, :E], o_true, labels[0], labels[1], 0, o)
        r_s, t_s = eng.rank_counts(both[:, E:], s_true, labels[2], labels[3], 0, s)
        for a, b in ((o_rank, r_o), (o_ties, t_o), (s_rank, r_s), (s_ties, t_s)):
            assert torch.equal(a, b), dt
        # raw + filtered from one scan per direction (ranges into a value array instead of a CSR)
        beg, end = labels[0][:-1].contiguous(), labels[0][1:].contiguous()
        cm = sh.rank_batch_multi(tri, [(beg, end, labels[1])], [(beg, end, labels[3])])
        raw = sh.rank_batch(tri, None)
        for k, (sr, st_, or_, ot) in enumerate((raw, (s_rank, s_ties, o_rank, o_ties))):
            assert torch.equal(cm[0, 0, k], or_) and torch.equal(cm[0, 1, k], ot), (dt, k)
            assert torch.equal(cm[1, 0, k], sr) and torch.equal(cm[1, 1, k], st_), (dt, k)
        tv, ti = sh.topk(sh.score_sp(s, p), 5)
        rv, ri = torch.topk(both[:, :E], 5, dim=1)
        assert torch.equal(tv, rv) and torch.equal(ti, ri)


@pytest.mark.parametrize("name", ["rotate", "transe", "complex", "distmult"])
def test_spo_backward_accumulated_with_repeated_indices(name):
    """Negative-sampling layout (s and p repeated K times in a row, random o with collisions, chunk
    boundaries inside runs): table gradients from the accumulate kernel vs torch autograd of the
    reference op sequence (oracle/torch_port.py on the GPU)."""
    from kge_amd import model as km
    E, R, d, n, K = 400, 5, 128, 37, 45
    g = torch.Generator().manual_seed(8)
    dr = d // 2 if name == "rotate" else d
    ent = (torch.randn(E, d, generator=g) * 0.5).to(DEV).requires_grad_(True)
    rel = (torch.randn(R, dr, generator=g) * 0.5).to(DEV).requires_grad_(True)
    s = torch.randint(E, (n,), generator=g).to(DEV).repeat_interleave(K)
    p = torch.randint(R, (n,), generator=g).to(DEV).repeat_interleave(K)
    o = torch.randint(E, (n * K,), generator=g).to(DEV)
    w = torch.randn(n * K, generator=g).to(DEV)
    (km._ScoreSPO.apply(name, 1.0, ent, rel, s, p, o) * w).sum().backward()
    ge, gr = ent.grad.clone(), rel.grad.clone()
    ent.grad = rel.grad = None
    (tp.score_emb(name, ent[s], rel[p], ent[o], "spo", 1.0).view(-1) * w).sum().backward()
    for got, want in ((ge, ent.grad), (gr, rel.grad)):
        torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-5 * float(want.abs().max()) + 1e-6)


@pytest.mark.parametrize("name", ["complex", "distmult"])
def test_sharded_ce_loss_on_the_gpu_kernels(name):
    """ShardedEntityTable.ce_loss on the real engine (one rank: the collectives degenerate, the dense-row
    entry points kge_ce_emb_fwd / kge_ce_emb_bwd and the scatter of the owned query rows do not): per-row
    losses and both table gradients against the index-level fused loss (kge_ce_fwd / kge_ce_bwd) on the
    same bf16 tables, and a few optimizer steps with float32 masters + refreshed bf16 scoring tables."""
    from kge_amd import engine as eng
    from kge_amd import model as km
    from kge_amd.optim import Adagrad
    from kge_amd.sharded import ShardedEntityTable
    E, R, d, n = 1500 + 7, 9, 256, 200
    g = torch.Generator().manual_seed(6)
    ent32 = (torch.randn(E, d, generator=g) * 0.3).to(DEV)
    rel32 = (torch.randn(R, d, generator=g) * 0.3).to(DEV)
    s, p, o = (torch.randint(hi, (n,), generator=g).to(DEV) for hi in (E, R, E))
    w = (torch.rand(2 * n, generator=g) + 0.5).to(DEV)
    ent_m, rel_m = ent32.clone().requires_grad_(True), rel32.clone().requires_grad_(True)
    sh = ShardedEntityTable(name, ent32.bfloat16(), rel32.bfloat16(), E)
    loss = torch.cat([sh.ce_loss("sp", s, p, o, ent_m, rel_m), sh.ce_loss("po", o, p, s, ent_m, rel_m)])
    (loss * w).sum().backward()
    # the unsharded fused loss on the same bf16 tables
    m = km.create(name, E, R, d, device=DEV, dtype=torch.bfloat16).train()
    with torch.no_grad():
        m.get_s_embedder().weight.copy_(ent32.bfloat16())
        m.get_p_embedder().weight.copy_(rel32.bfloat16())
    T = eng.Tables(name, m.get_s_embedder().weight.detach(), m.get_p_embedder().weight.detach())
    l_sp, lse_sp = eng.ce_fwd(T, "sp", s, p, o)
    l_po, lse_po = eng.ce_fwd(T, "po", o, p, s)
    torch.testing.assert_close(loss, torch.cat([l_sp, l_po]), rtol=1e-5, atol=1e-5)
    ge, gr = torch.zeros(E, d, device=DEV), torch.zeros(R, d, device=DEV)
    for direction, a, lab, lse, ww in (("sp", s, o, lse_sp, w[:n]), ("po", o, s, lse_po, w[n:])):
        g_a, g_p, g_t = eng.ce_bwd(T, direction, a, p, lab, lse, g_rows=ww)
        ge += g_t
        ge.index_add_(0, a, g_a)
        gr.index_add_(0, p, g_p)
    for got, want in ((ent_m.grad, ge), (rel_m.grad, gr)):
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4 * float(want.abs().max()))
    # three training steps: float32 masters, bf16 scoring tables refreshed after every step; the loss falls
    opt = Adagrad([ent_m, rel_m], lr=0.1)
    first = last = None
    for step in range(3):
        opt.zero_grad()
        rows = torch.cat([sh.ce_loss("sp", s, p, o, ent_m, rel_m), sh.ce_loss("po", o, p, s, ent_m, rel_m)])
        val = rows.sum() / n
        val.backward()
        opt.step()
        sh.refresh_tables(ent_m, rel_m)
        first = float(val.detach()) if first is None else first
        last = float(val.detach())
    assert last < first


def test_sharded_path_through_a_real_rccl_process_group():
    """The sharded exchange with REAL collectives on the GPU: a one-rank process group on the nccl (= RCCL)
    backend and `force_collectives=True`, so that all_gather_into_tensor / all_reduce / all_gather run through
    RCCL on device buffers instead of being skipped as identities ("the real engine has never met
    a real collective in a test"; a 1-GPU box cannot host two ranks).  Results must equal the same table
    without a process group: ranks, top-k, and the sharded 1vsAll loss with both table gradients."""
    import torch.distributed as dist
    from kge_amd.sharded import ShardedEntityTable
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    g = torch.Generator().manual_seed(8)
    E, R, d, n = 900, 5, 256, 64
    ent32 = (torch.randn(E, d, generator=g) * 0.3).to(DEV)
    rel32 = (torch.randn(R, d, generator=g) * 0.3).to(DEV)
    tri = torch.stack([torch.randint(E, (n,), generator=g), torch.randint(R, (n,), generator=g),
                       torch.randint(E, (n,), generator=g)], 1).to(DEV)
    s, p, o = tri[:, 0], tri[:, 1], tri[:, 2]

    def run(sh):
        ranks = sh.rank_batch(tri, None)
        slab = sh.score_sp(s, p)
        top = sh.topk(slab, 5)
        ent_m, rel_m = ent32.clone().requires_grad_(True), rel32.clone().requires_grad_(True)
        loss = sh.ce_loss("sp", s, p, o, ent_m, rel_m)
        loss.sum().backward()
        return ranks, top, loss.detach(), ent_m.grad, rel_m.grad

    plain = run(ShardedEntityTable("complex", ent32.bfloat16(), rel32.bfloat16(), E))
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29547", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        sh = ShardedEntityTable("complex", ent32.bfloat16(), rel32.bfloat16(), E, force_collectives=True)
        assert sh.collectives and sh.world == 1
        forced = run(sh)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    for a, b in zip(plain[0], forced[0]):
        assert torch.equal(a, b)
    assert torch.equal(plain[1][0], forced[1][0]) and torch.equal(plain[1][1], forced[1][1])
    torch.testing.assert_close(plain[2], forced[2], rtol=1e-6, atol=1e-7)   # the per-row losses
    for a, b in zip(plain[3:], forced[3:]):  # gradients: float atomics / index_add accumulate in any order
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)
