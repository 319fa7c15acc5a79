"""kge_amd.train_graph.GraphedStep: the 1vsAll training step (fused cross entropy of both directions, its backward, the
one-pass Adagrad) replayed as one hipGraph takes the same steps as the eager loop -- the reference's
zero_grad / forward / backward / optimizer.step (kge/job/train.py:452-474).

Bar: the same kernels on the same inputs; the only freedom is the order of the float atomics that scatter the query
rows' gradients (index_add_): losses to 5e-5 relative over a dozen steps, SGD parameters to 1e-4."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(graphed: bool, batches, lr_change_at=None, optimizer="Adagrad"):
    from kge_amd import model as km, optim as kopt
    from kge_amd.train_graph import GraphedStep
    torch.manual_seed(0)
    m = km.create("complex", 3001, 7, 256, device=DEV, score_dtype=torch.bfloat16)
    if optimizer == "Adagrad":
        opt = kopt.Adagrad(m.parameters(), lr=0.1, bf16_copies=True)
    else:
        opt = torch.optim.SGD(m.parameters(), lr=0.1)
    step = GraphedStep(lambda s, p, o: m.loss_sp_po(s, p, o).sum() / len(s), opt, warmup=2, enabled=graphed)
    losses = []
    for k, b in enumerate(batches):
        if lr_change_at is not None and k == lr_change_at:
            for g in opt.param_groups:
                g["lr"] = 0.05
        losses.append(float(step(b[:, 0], b[:, 1], b[:, 2])))
    return losses, [p.detach().clone() for p in m.parameters()], step


def test_graphed_step_takes_the_eager_steps():
    g = torch.Generator().manual_seed(1)
    mk = lambda n: torch.stack([torch.randint(hi, (n,), generator=g) for hi in (3001, 7, 3001)], 1).to(DEV)
    batches = [mk(256) for _ in range(6)] + [mk(100)] + [mk(256) for _ in range(4)]   # a short batch in between
    for optimizer in ("Adagrad", "SGD"):
        l_e, p_e, _ = _run(False, batches, lr_change_at=8, optimizer=optimizer)
        l_g, p_g, step = _run(True, batches, lr_change_at=8, optimizer=optimizer)
        assert step.disabled_reason is None
        assert step.replays == len(batches) - 2 - 1      # all but the warm-up steps and the short batch
        assert step.captures == 2                        # the first capture and the one behind the learning-rate change
        # (a replay orders the float atomics of the gradient scatter differently from the eager launches; over a dozen
        # steps that is a few 1e-6 relative in the loss -- the replay bug this module once had was 1e-3 and growing)
        for a, b in zip(l_e, l_g):
            assert abs(a - b) <= 5e-5 * abs(a) + 1e-7, (optimizer, l_e, l_g)
        assert l_e[0] > l_e[-1]                          # it trains
        if optimizer == "SGD":
            # (Adagrad's first steps are lr * g / |g|: a coordinate whose gradient is atomics noise around zero moves
            # by +-lr in either run -- its eleven losses above agree to 1e-6, its parameters are not compared)
            for a, b in zip(p_e, p_g):
                torch.testing.assert_close(a, b, rtol=0, atol=1e-4)


def test_graphed_step_refuses_a_step_count_dependent_optimizer():
    from kge_amd import model as km, optim as kopt
    from kge_amd.train_graph import GraphedStep
    m = km.create("distmult", 500, 3, 256, device=DEV, score_dtype=torch.bfloat16)
    opt = kopt.Adagrad(m.parameters(), lr=0.1, lr_decay=0.01, bf16_copies=True)
    step = GraphedStep(lambda s, p, o: m.loss_sp_po(s, p, o).sum(), opt, warmup=0)
    assert not step.enabled and "lr_decay" in step.disabled_reason
    b = torch.stack([torch.randint(hi, (64,)) for hi in (500, 3, 500)], 1).to(DEV)
    for _ in range(3):
        step(b[:, 0], b[:, 1], b[:, 2])
    assert step.replays == 0


def test_replayed_gradients_stay_the_eager_gradients_over_many_replays():
    """260 replays of a captured step with device allocations of assorted sizes, pageable host <-> device copies and a
    loss read-back in between; from the 100th on the captured step's gradient buffers (GraphedStep.static_grads) are
    compared with an eager step from the same parameters.
    Background: round 4's replay bug -- a hipMemsetAsync captured into a hipGraph is replayed from a pattern the graph
    does not own; under LibKGE's trainer, from the ~100th replay on, the relation-gradient accumulator the library
    "zeroed" came back as a repeating 16-byte pattern with a garbage dword.  The library fills with its own kernel now
    (common.hpp fill_words_async).  THIS test did not reproduce the failure with the old library (the trigger needs
    more of LibKGE's host-side activity than is imitated here); the test that did, deterministically, is plugin test
    `a` in tests/test_gpu_libkge_plugin.py (second epoch), which needs the reference package on the box."""
    from kge_amd import model as km
    from kge_amd.train_graph import GraphedStep
    E, R, D, N = 3001, 237, 512, 256
    torch.manual_seed(0)
    m_g = km.create("complex", E, R, D, device=DEV, score_dtype=torch.bfloat16)
    m_e = km.create("complex", E, R, D, device=DEV, score_dtype=torch.bfloat16)
    opt_g = torch.optim.SGD(m_g.parameters(), lr=0.05)
    step = GraphedStep(lambda s, p, o: m_g.loss_sp_po(s, p, o).sum() / len(s), opt_g, warmup=2)
    g = torch.Generator().manual_seed(9)
    churn = []
    for k in range(260):
        b = torch.stack([torch.randint(hi, (N,), generator=g) for hi in (E, R, E)], 1).to(DEV)
        if k >= 100:
            m_e.load_state_dict(m_g.state_dict())
            for p in m_e.parameters():
                p.grad = None
            (m_e.loss_sp_po(b[:, 0], b[:, 1], b[:, 2]).sum() / N).backward()
        loss = float(step(b[:, 0], b[:, 1], b[:, 2]))          # (a read-back per step, as LibKGE's trainer does)
        assert loss == loss
        if k >= 100:
            assert step.replays == k - 2 + 1 and len(step.static_grads) == 2
            for got, p in zip(step.static_grads, m_e.parameters()):
                err = float((got - p.grad).abs().max())
                assert err <= 1e-6 * float(p.grad.abs().max()) + 1e-9, (k, tuple(got.shape), err, float(got.abs().max()))
        # churn between the replays: what a training loop's own temporaries and copies do (device blocks of assorted
        # sizes; pageable host <-> device copies, which go through the runtime's staging buffers)
        host = torch.randn(40000 + 997 * (k % 7))
        dev_copy = host.to(DEV)
        back = (dev_copy * 2).cpu()
        assert float(back[0]) == float(host[0]) * 2
        churn.append([torch.empty(sz, dtype=torch.uint8, device=DEV).fill_(0xAB) for sz in (512, 12288, 485376, 1 << 20, 7 << 20)])
        if len(churn) > 3:
            churn.pop(0)


def test_graphed_step_refuses_adam_and_counts_adagrad_steps():
    """kge_amd.optim.Adam computes lr / (1 - beta1^t) and sqrt(1 - beta2^t) on the host and hands them to the
    kernel as launch arguments -- a capture would freeze them at the capture step.  GraphedStep must refuse it (the step
    stays eager and equals an eager run bit for bit up to the scatter atomics), and for Adagrad the per-parameter step
    count of the checkpoint (state["step"]) must keep counting through replays."""
    from kge_amd import model as km, optim as kopt
    from kge_amd.train_graph import GraphedStep
    g = torch.Generator().manual_seed(3)
    batches = [torch.stack([torch.randint(hi, (128,), generator=g) for hi in (900, 5, 900)], 1).to(DEV) for _ in range(8)]

    def run(graphed):
        torch.manual_seed(0)
        m = km.create("complex", 900, 5, 256, device=DEV, score_dtype=torch.bfloat16)
        opt = kopt.Adam(m.parameters(), lr=0.01, bf16_copies=True)
        step = GraphedStep(lambda s, p, o: m.loss_sp_po(s, p, o).sum() / len(s), opt, warmup=2, enabled=graphed)
        return [float(step(b[:, 0], b[:, 1], b[:, 2])) for b in batches], opt, step
    l_e, _, _ = run(False)
    l_g, opt, step = run(True)
    assert not step.enabled and "step count" in step.disabled_reason and step.replays == 0
    for a, b in zip(l_e, l_g):
        assert abs(a - b) <= 5e-5 * abs(a) + 1e-7
    assert all(float(st["step"]) == len(batches) for st in opt.state.values())

    torch.manual_seed(0)
    m = km.create("complex", 900, 5, 256, device=DEV, score_dtype=torch.bfloat16)
    opt = kopt.Adagrad(m.parameters(), lr=0.1, bf16_copies=True)
    step = GraphedStep(lambda s, p, o: m.loss_sp_po(s, p, o).sum() / len(s), opt, warmup=2)
    for b in batches:
        step(b[:, 0], b[:, 1], b[:, 2])
    assert step.replays == len(batches) - 2 and step.disabled_reason is None
    assert all(float(st["step"]) == len(batches) for st in opt.state.values()), [float(st["step"]) for st in opt.state.values()]
