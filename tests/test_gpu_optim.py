"""kge_amd.optim.Adagrad (kge_adagrad_step: one pass per table) against torch.optim.Adagrad:
same trajectory within float rounding (torch's element-wise kernels may contract `sum + g*g` and
`g + wd*p` into fma, this kernel rounds every operation: a few ulp per step), interchangeable
state_dicts, and the bf16 copies of the tables written in the same pass."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("lr_decay,weight_decay,init_acc", [(0.0, 0.0, 0.0), (0.01, 1e-3, 0.1)])
def test_adagrad_matches_torch(lr_decay, weight_decay, init_acc):
    from kge_amd.optim import Adagrad
    torch.manual_seed(0)
    shapes = [(1001, 96), (7, 33), (5,)]     # a table, an odd-sized one (scalar tail), a vector
    ref = [torch.randn(s, device=DEV).requires_grad_(True) for s in shapes]
    got = [r.detach().clone().requires_grad_(True) for r in ref]
    o_ref = torch.optim.Adagrad(ref, lr=0.1, lr_decay=lr_decay, weight_decay=weight_decay,
                                initial_accumulator_value=init_acc, eps=1e-10)
    o_got = Adagrad(got, lr=0.1, lr_decay=lr_decay, weight_decay=weight_decay,
                    initial_accumulator_value=init_acc, eps=1e-10, bf16_copies=True)
    for step in range(4):
        for r, g in zip(ref, got):
            grad = torch.randn_like(r)
            r.grad, g.grad = grad.clone(), grad.clone()
        v0 = [g._version for g in got]
        o_ref.step()
        o_got.step()
        for r, g, v in zip(ref, got, v0):
            torch.testing.assert_close(g.detach(), r.detach(), rtol=2e-6, atol=1e-7)
            assert g._version > v                       # the raw-pointer update is visible to autograd
        for (kr, sr), (kg, sg) in zip(o_ref.state.items(), o_got.state.items()):
            torch.testing.assert_close(sg["sum"], sr["sum"], rtol=2e-6, atol=1e-30)
            assert float(sg["step"]) == float(sr["step"]) == step + 1
    # bf16 copies: only for 2-D parameters, equal to the RNE cast of the new parameter, fresh
    from kge_amd.optim import bf16_copy_of
    for g in got:
        c = bf16_copy_of(g)
        if g.dim() == 2:
            assert c is not None and torch.equal(c, g.detach().to(torch.bfloat16))
        else:
            assert c is None
    got[0].data.mul_(1.0)            # .data edits do not bump the version ...
    with torch.no_grad():
        got[0].mul_(1.0)             # ... in-place ops do: the copy is stale now
    assert bf16_copy_of(got[0]) is None
    # state_dicts are interchangeable
    o_ref.load_state_dict(o_got.state_dict())
    o_got.load_state_dict(o_ref.state_dict())


def test_adagrad_multi_is_the_single_launches_bit_for_bit():
    """kge_adagrad_step_multi (all dense tables of a step in one launch) against one kge_adagrad_step per table:
    the same bits in parameters, accumulators and bf16 copies; eleven tables = two launches of the optimizer (eight
    segments per launch), sizes with scalar tails and an empty one."""
    import ctypes
    from kge_amd import _lib
    from kge_amd.engine import _stream
    torch.manual_seed(1)
    sizes = [4096 * 3 + 1, 5, 0, 1024, 1023, 1025, 231, 64 * 512, 8, 12, 100003]
    mk = lambda: [torch.randn(max(k, 1), device=DEV)[:k] for k in sizes]
    p0, g0, s0 = mk(), mk(), [x.abs() for x in mk()]
    res = {}
    for mode in ("single", "multi"):
        ps, ss = [x.clone() for x in p0], [x.clone() for x in s0]
        cs = [torch.zeros(k, dtype=torch.bfloat16, device=DEV) if i % 2 == 0 else None for i, k in enumerate(sizes)]
        ptr = lambda t: None if t is None or t.numel() == 0 else t.data_ptr()
        if mode == "single":
            for p_, g_, s_, c_ in zip(ps, g0, ss, cs):
                _lib.check(_lib.lib().kge_adagrad_step(ptr(p_), ptr(g_), ptr(s_), p_.numel(), -0.05, 1e-3, 1e-10,
                                                       ptr(c_), _stream(torch.device(DEV))), "kge_adagrad_step")
        else:
            segs = [_lib.KgeAdagradSeg(ptr(p_), ptr(g_), ptr(s_), ptr(c_), p_.numel(), -0.05, 1e-3, 1e-10)
                    for p_, g_, s_, c_ in zip(ps, g0, ss, cs)]
            assert _lib.lib().kge_adagrad_step_multi((_lib.KgeAdagradSeg * 9)(*segs[:9]), 9, None) == -1  # > 8 segments
            for i in range(0, len(segs), 8):
                chunk = segs[i:i + 8]
                _lib.check(_lib.lib().kge_adagrad_step_multi((_lib.KgeAdagradSeg * len(chunk))(*chunk), len(chunk),
                                                             _stream(torch.device(DEV))), "kge_adagrad_step_multi")
        torch.cuda.synchronize()
        res[mode] = (ps, ss, cs)
    for a, b in zip(res["single"], res["multi"]):
        for x, y in zip(a, b):
            assert (x is None and y is None) or torch.equal(x, y)
    assert not torch.equal(res["multi"][0][0], p0[0])


def test_mixed_precision_model_uses_the_optimizer_copies():
    """score_dtype=bfloat16 + Adagrad(bf16_copies=True): after a step the scoring tables ARE the
    optimizer's copies (no cast kernels), and the scores equal those of freshly cast tables."""
    from kge_amd import engine as eng
    from kge_amd import model as km
    from kge_amd.optim import Adagrad, bf16_copy_of
    E, R, d, n = 1500, 9, 256, 130
    torch.manual_seed(0)
    m = km.create("complex", E, R, d, device=DEV, score_dtype=torch.bfloat16)
    opt = Adagrad(m.parameters(), lr=0.1, bf16_copies=True)
    g = torch.Generator().manual_seed(1)
    s, p, o = (torch.randint(hi, (n,), generator=g).to(DEV) for hi in (E, R, E))
    for _ in range(2):
        opt.zero_grad()
        (m.loss_sp(s, p, o).sum() / n).backward()
        opt.step()
    ent, rel = m._entity_embedder.weight, m._relation_embedder.weight
    t = m._fwd_tables()
    assert t.ent.data_ptr() == bf16_copy_of(ent).data_ptr() and t.rel.data_ptr() == bf16_copy_of(rel).data_ptr()
    fresh = eng.Tables("complex", ent.detach().bfloat16(), rel.detach().bfloat16())
    with torch.no_grad():
        assert torch.equal(m.score_sp(s, p), eng.score_sp(fresh, s, p))


def test_whole_training_step_replayed_as_a_hip_graph():
    """Fused loss (both directions) + backward + one-pass Adagrad captured ONCE into a hipGraph
    (after an eager warm-up, which tunes the GEMM plans) and replayed: parameters after k replays
    equal those of k eager steps from the same state (same kernels, same plans; the captured
    scoring launches build their query fragments per workgroup instead of cooperatively --
    identical bits)."""
    from kge_amd import model as km
    from kge_amd.optim import Adagrad
    E, R, d, n = 2000 + 7, 11, 256, 256
    g = torch.Generator().manual_seed(3)
    s, p, o = (torch.randint(hi, (n,), generator=g).to(DEV) for hi in (E, R, E))

    def make():
        torch.manual_seed(0)
        m = km.create("distmult", E, R, d, device=DEV, score_dtype=torch.bfloat16)
        return m, Adagrad(m.parameters(), lr=0.05, bf16_copies=True)

    def step(m, opt):
        opt.zero_grad(set_to_none=True)
        (m.loss_sp(s, p, o).sum() / n).backward()
        (m.loss_po(p, o, s).sum() / n).backward()
        opt.step()

    m_e, o_e = make()
    m_g, o_g = make()
    warm, k = 2, 3
    for _ in range(warm + k):
        step(m_e, o_e)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            step(m_g, o_g)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    o_g.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        (m_g.loss_sp(s, p, o).sum() / n).backward()
        (m_g.loss_po(p, o, s).sum() / n).backward()
        o_g.step()
    # the capture itself executes nothing: k replays = k steps
    for _ in range(k):
        graph.replay()
    torch.cuda.synchronize()
    # (the same kernels on the same inputs: what differs between a replay and the eager launches is the order of the
    # float atomics that scatter the query rows' gradients -- a few 1e-7 after k steps; one element in 2,816 was seen
    # at 4e-7)
    for a, b in zip(m_e.parameters(), m_g.parameters()):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("weight_decay", [0.0, 1e-3])
def test_adam_matches_torch(weight_decay):
    """kge_amd.optim.Adam (kge_adam_step: one pass per table) against torch.optim.Adam over 5 steps:
    parameters and both moment estimates within float rounding, interchangeable state_dicts, bf16
    copies written in the same pass; a CPU parameter in the same optimizer is stepped by torch."""
    from kge_amd.optim import Adam, bf16_copy_of
    torch.manual_seed(1)
    shapes = [(777, 64), (9, 31), (6,)]
    ref = [torch.randn(s, device=DEV).requires_grad_(True) for s in shapes] + [torch.randn(4, 3, requires_grad=True)]
    got = [r.detach().clone().requires_grad_(True) for r in ref]
    o_ref = torch.optim.Adam(ref, lr=0.05, betas=(0.9, 0.99), eps=1e-8, weight_decay=weight_decay)
    o_got = Adam(got, lr=0.05, betas=(0.9, 0.99), eps=1e-8, weight_decay=weight_decay, bf16_copies=True)
    for step in range(5):
        for r, g in zip(ref, got):
            grad = torch.randn_like(r)
            r.grad, g.grad = grad.clone(), grad.clone()
        o_ref.step()
        o_got.step()
        for r, g in zip(ref, got):
            torch.testing.assert_close(g.detach(), r.detach(), rtol=5e-6, atol=2e-6)
        for sr, sg in zip(o_ref.state.values(), o_got.state.values()):
            torch.testing.assert_close(sg["exp_avg"], sr["exp_avg"], rtol=5e-6, atol=2e-7)
            torch.testing.assert_close(sg["exp_avg_sq"], sr["exp_avg_sq"], rtol=5e-6, atol=1e-9)
            assert float(sg["step"]) == float(sr["step"]) == step + 1
    c = bf16_copy_of(got[0])
    assert c is not None and torch.equal(c, got[0].detach().to(torch.bfloat16))
    o_ref.load_state_dict(o_got.state_dict())
    o_got.load_state_dict(o_ref.state_dict())


def test_adagrad_row_sparse_gradients_match_torch():
    """lookup_embedder.sparse: True gives nn.Embedding(sparse=True) and with it row-sparse gradients:
    kge_adagrad_step_rows (only the touched rows are read and written) against torch.optim.Adagrad's
    sparse path, duplicate ids in the batch included; the bf16 copy follows the touched rows."""
    from kge_amd.optim import Adagrad, bf16_copy_of
    torch.manual_seed(2)
    E, d = 5000, 96
    w0 = torch.randn(E, d, device=DEV)
    emb_ref = torch.nn.Embedding(E, d, sparse=True, device=DEV)
    emb_got = torch.nn.Embedding(E, d, sparse=True, device=DEV)
    with torch.no_grad():
        emb_ref.weight.copy_(w0)
        emb_got.weight.copy_(w0)
    o_ref = torch.optim.Adagrad(emb_ref.parameters(), lr=0.1, lr_decay=0.01, eps=1e-10)
    o_got = Adagrad(emb_got.parameters(), lr=0.1, lr_decay=0.01, eps=1e-10, bf16_copies=True)
    for step in range(4):
        idx = torch.randint(E, (300,), device=DEV)
        idx[:20] = idx[20:40]  # duplicates: coalesced before the update
        wgt = torch.randn(300, d, device=DEV)
        for emb, opt in ((emb_ref, o_ref), (emb_got, o_got)):
            opt.zero_grad()
            (emb(idx) * wgt).sum().backward()
            assert emb.weight.grad.is_sparse
            opt.step()
        torch.testing.assert_close(emb_got.weight.detach(), emb_ref.weight.detach(), rtol=2e-6, atol=1e-7)
        torch.testing.assert_close(o_got.state[emb_got.weight]["sum"], o_ref.state[emb_ref.weight]["sum"],
                                   rtol=2e-6, atol=1e-30)
    c = bf16_copy_of(emb_got.weight)
    assert c is not None and torch.equal(c, emb_got.weight.detach().to(torch.bfloat16))


def test_adagrad_row_sparse_step_refreshes_a_stale_bf16_copy():
    """A parameter changed behind the optimizer's back (checkpoint loaded into the same Parameter, re-init, a dense
    torch step) makes the bf16 copy stale; the row-sparse step rewrites only the touched rows, so it must re-cast
    the copy first -- otherwise every other row would score with the OLD embeddings."""
    from kge_amd.optim import Adagrad, bf16_copy_of
    torch.manual_seed(5)
    E, d = 800, 64
    emb = torch.nn.Embedding(E, d, sparse=True, device=DEV)
    opt = Adagrad(emb.parameters(), lr=0.1, bf16_copies=True)

    def sparse_step():
        idx = torch.randint(E, (50,), device=DEV)
        opt.zero_grad()
        emb(idx).sum().backward()
        opt.step()

    sparse_step()
    assert torch.equal(bf16_copy_of(emb.weight), emb.weight.detach().to(torch.bfloat16))
    with torch.no_grad():  # the whole table changes outside the optimizer
        emb.weight.copy_(torch.randn(E, d, device=DEV))
    assert bf16_copy_of(emb.weight) is None
    sparse_step()
    c = bf16_copy_of(emb.weight)
    assert c is not None and torch.equal(c, emb.weight.detach().to(torch.bfloat16))


@pytest.mark.parametrize("kind,p", [("lp", 2), ("lp", 1), ("lp", 3), ("n3_complex", 3)])
def test_adagrad_with_folded_penalty_equals_backpropagating_the_reference_term(kind, p):
    """kge_adagrad_step_multi_penalty: the unweighted penalty of LookupEmbedder.penalty
    (kge/model/embedder/lookup_embedder.py:122-147: weight / p * norm(p) ** p, n3 over sqrt(re^2 + im^2 + 1e-14); the entity
    table's term doubled, kge_model.py:620-625) added to the gradient inside the update's pass == torch autograd of the
    term followed by the plain step; the value the pass sums == the term (of the pre-step parameters)."""
    from kge_amd.optim import Adagrad
    torch.manual_seed(3)
    shapes = [(1003, 64), (37, 64)]      # "entity" table (times 2), "relation" table; the first has a scalar tail for lp
    weights = [0.013, 0.4]
    times = [2.0, 1.0]
    ref = [torch.randn(s, device=DEV).requires_grad_(True) for s in shapes]
    with torch.no_grad():
        ref[0][5, :8] = 0.0              # exact zeros: sign(0) = 0, |z| = 1e-7
    got = [r.detach().clone().requires_grad_(True) for r in ref]
    o_ref = Adagrad(ref, lr=0.1, bf16_copies=True)
    o_got = Adagrad(got, lr=0.1, bf16_copies=True)
    for g, w, t in zip(got, weights, times):
        o_got.set_penalty(g, kind, p, w, t)

    def term(x, w, t):
        if kind == "n3_complex":
            re, im = (c.contiguous() for c in x.chunk(2, dim=1))
            x = torch.sqrt(re ** 2 + im ** 2 + 1e-14)
        return (w / p * x.norm(p=p) ** p).sum() * t
    for step in range(3):
        for r, g in zip(ref, got):
            grad = torch.randn_like(r) * 0.1
            r.grad, g.grad = grad.clone(), grad.clone()
        values = []
        for r, w, t in zip(ref, weights, times):
            v = term(r, w, t)
            v.backward()
            values.append(float(v))
        o_ref.step()
        o_got.step()
        for r, g, v in zip(ref, got, values):
            torch.testing.assert_close(g.detach(), r.detach(), rtol=1e-5, atol=2e-6)
            assert abs(float(o_got.penalty_value(g)) - v) <= 2e-6 * abs(v)
        for sr, sg in zip(o_ref.state.values(), o_got.state.values()):
            torch.testing.assert_close(sg["sum"], sr["sum"], rtol=1e-5, atol=1e-9)
    # the plain step on request (the caller back-propagated the term itself), the terms again afterwards
    for r, g in zip(ref, got):
        grad = torch.randn_like(r) * 0.1
        r.grad, g.grad = grad.clone(), grad.clone()
    o_got.skip_penalties_once()
    o_ref.step()
    o_got.step()
    for r, g in zip(ref, got):
        torch.testing.assert_close(g.detach(), r.detach(), rtol=1e-5, atol=2e-6)
    assert o_got._pen_off_once is False
