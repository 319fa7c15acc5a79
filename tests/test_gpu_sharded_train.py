"""ShardedTrainingJob1vsAll on the HIP engine (one rank; the two-rank choreography is tests/test_sharded_gloo_cpu.py
and tests/test_gpu_sharded_two_ranks.py): three 1vsAll optimizer steps with the fused score + loss kernels on bf16
scoring copies against the reference's step in float32 torch ops on the same bf16-rounded tables
(train_1vsAll.py:64-81 with torch.optim.Adagrad), and the checkpoint round trip.

Bar: mixed precision -- scores from bf16 tables with a bf16-rounded query vector, d loss / d score rounded to bf16 for
the gradient products: losses within 2e-3 relative, parameters after three steps within 2e-3 of the
step size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("model,d", [("complex", 256), ("distmult", 512)])
def test_sharded_job_on_the_engine_follows_the_reference_step(model, d):
    import torch.nn.functional as F
    import torch_port as tp
    from kge_amd.sharded_train import ENT_KEY, REL_KEY, ShardedTrainingJob1vsAll
    E, R, n, lr = 3001, 7, 128, 0.5
    g = torch.Generator().manual_seed(5)
    batches = [torch.stack([torch.randint(hi, (n,), generator=g) for hi in (E, R, E)], 1) for _ in range(3)]
    # plain SGD for the comparison of the parameters (Adagrad's first steps are lr * sign(g): a gradient coordinate
    # near zero may change sign under bf16 rounding and move by 2 lr)
    job = ShardedTrainingJob1vsAll(model, E, R, d, seed=11, lr=lr, optimizer="SGD", device=DEV)
    sd0 = job.state_dict()
    ent = sd0[ENT_KEY].clone().requires_grad_(True)
    rel = sd0[REL_KEY].clone().requires_grad_(True)
    opt = torch.optim.SGD([ent, rel], lr=lr)
    for k, b in enumerate(batches):
        got = float(job.step(b))
        s, p, o = b[:, 0], b[:, 1], b[:, 2]
        opt.zero_grad()
        e16 = ent + (ent.detach().bfloat16().float() - ent.detach())   # straight-through: bf16 values, f32 gradients
        r16 = rel + (rel.detach().bfloat16().float() - rel.detach())
        want = (F.cross_entropy(tp.score_sp(model, e16, r16, s, p), o, reduction="sum") +
                F.cross_entropy(tp.score_po(model, e16, r16, p, o), s, reduction="sum")) / n
        want.backward()
        opt.step()
        assert abs(got - float(want.detach())) <= 2e-3 * abs(float(want.detach())), (k, got, float(want.detach()))
    sd = job.state_dict()
    assert sd[ENT_KEY].shape == (E, d)
    moved = float((ent.detach() - sd0[ENT_KEY]).abs().max())
    assert moved > 1e-4
    assert float((sd[ENT_KEY] - ent.detach()).abs().max()) <= 1e-2 * moved + 1e-5
    assert float((sd[REL_KEY] - rel.detach()).abs().max()) <= 1e-2 * float((rel.detach() - sd0[REL_KEY]).abs().max()) + 1e-5
    # the one-pass Adagrad kernel on this rank's rows + the checkpoint round trip: a fresh job resumed from the
    # checkpoint takes the same next step
    job = ShardedTrainingJob1vsAll(model, E, R, d, seed=11, lr=0.1, optimizer="Adagrad", device=DEV)
    assert type(job.optimizer).__module__ == "kge_amd.optim"
    for b in batches[:2]:
        job.step(b)
    ck = job.checkpoint()
    assert ck["model"][1][ENT_KEY].shape == (E, d) and ck["optimizer_state"][ENT_KEY]["sum"].shape == (E, d)
    l_a = float(job.step(batches[2]))
    job2 = ShardedTrainingJob1vsAll(model, E, R, d, seed=12, lr=0.1, optimizer="Adagrad", device=DEV)
    job2.load_checkpoint(ck)
    l_b = float(job2.step(batches[2]))
    assert l_a == l_b   # same tables, same optimizer state: the same forward
    # (the query-row gradients are scatter-added with float atomics -- index_add_ -- whose order is not fixed)
    torch.testing.assert_close(job.state_dict()[ENT_KEY], job2.state_dict()[ENT_KEY], rtol=0, atol=1e-5)
