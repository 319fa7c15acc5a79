"""ShardedTrainingJob1vsAll on the HIP engine (one rank; the two-rank choreography is tests/test_sharded_gloo_cpu.py
and tests/test_gpu_sharded_two_ranks.py): three 1vsAll optimizer steps with the fused score + loss kernels on bf16
scoring copies against the reference's step in float32 torch ops on the same bf16-rounded tables
(train_1vsAll.py:64-81 with torch.optim.Adagrad), and the checkpoint round trip.

Bar: mixed precision -- scores from bf16 tables with a bf16-rounded query vector, d loss / d score rounded to bf16 for
the gradient products: losses within 2e-3 relative, parameters after three steps within 2e-3 of the
step size."""
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
    assert ck["model"][0][ENT_KEY].shape == (E, d) and ck["optimizer_state_dict"]["state"][0]["sum"].shape == (E, d)
    assert ck["type"] == "train" and "valid_trace" in ck   # TrainingJob.save_to's keys (kge/job/train.py:284-298)
    l_a = float(job.step(batches[2]))
    job2 = ShardedTrainingJob1vsAll(model, E, R, d, seed=12, lr=0.1, optimizer="Adagrad", device=DEV)
    job2.load_checkpoint(ck)
    l_b = float(job2.step(batches[2]))
    assert l_a == l_b   # same tables, same optimizer state: the same forward
    # (the query-row gradients are scatter-added with float atomics -- index_add_ -- whose order is not fixed)
    torch.testing.assert_close(job.state_dict()[ENT_KEY], job2.state_dict()[ENT_KEY], rtol=0, atol=1e-5)


def _labels(g, n, E, kmax=6):
    cnt = torch.randint(0, kmax + 1, (n,), generator=g)
    cnt[0] = 0
    rowptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(cnt, 0)])
    col = torch.cat([torch.randperm(E, generator=g)[:int(c)] for c in cnt] + [torch.zeros(0, dtype=torch.int64)])
    return rowptr, col


@pytest.mark.parametrize("model,d,loss", [("complex", 256, "kl"), ("distmult", 512, "kl"), ("complex", 512, "bce")])
def test_sharded_kvsall_job_on_the_engine(model, d, loss):
    """ShardedTrainingJobKvsAll on the HIP engine (one rank: kge_kl_weighted_emb_* / kge_bce_emb_* on dense query rows,
    labels as GLOBAL ids against the shard): three SGD steps against TrainingJobKvsAll's step in float32 torch ops on the
    same bf16-rounded tables (train_KvsAll.py:274-294).  Mixed-precision bar as above."""
    import torch.nn.functional as F
    import torch_port as tp
    from kge_amd.sharded_train import ENT_KEY, REL_KEY, ShardedTrainingJobKvsAll
    E, R, n, lr = 3001, 7, 96, 0.5
    g = torch.Generator().manual_seed(7)
    job = ShardedTrainingJobKvsAll(model, E, R, d, seed=13, lr=lr, optimizer="SGD", device=DEV, loss=loss, loss_arg=0.1)
    sd0 = job.state_dict()
    ent = sd0[ENT_KEY].clone().requires_grad_(True)
    rel = sd0[REL_KEY].clone().requires_grad_(True)
    opt = torch.optim.SGD([ent, rel], lr=lr)
    for k in range(3):
        groups = []
        for direction in ("sp", "po"):
            ids, p = torch.randint(E, (n,), generator=g), torch.randint(R, (n,), generator=g)
            groups.append((direction, ids, p) + _labels(g, n, E))
        got = float(job.step(groups))
        opt.zero_grad()
        e16 = ent + (ent.detach().bfloat16().float() - ent.detach())
        r16 = rel + (rel.detach().bfloat16().float() - rel.detach())
        want = 0.0
        for direction, ids, p, rowptr, col in groups:
            sc = tp.score_sp(model, e16, r16, ids, p) if direction == "sp" else tp.score_po(model, e16, r16, p, ids)
            y = torch.zeros(n, E)
            for i in range(n):
                y[i, col[rowptr[i]:rowptr[i + 1]]] = 1.0
            if loss == "kl":
                l = F.kl_div(F.log_softmax(sc, dim=1), F.normalize(y, p=1, dim=1), reduction="sum") / (2 * n)
            else:
                l = F.binary_cross_entropy_with_logits(sc + 0.1, y, reduction="sum") / (2 * n)
            l.backward()
            want += float(l.detach())
        opt.step()
        assert abs(got - want) <= 2e-3 * abs(want), (k, got, want)
    sd = job.state_dict()
    moved = float((ent.detach() - sd0[ENT_KEY]).abs().max())
    assert moved > 1e-5
    assert float((sd[ENT_KEY] - ent.detach()).abs().max()) <= 1e-2 * moved + 1e-5
    assert float((sd[REL_KEY] - rel.detach()).abs().max()) <= 1e-2 * float((rel.detach() - sd0[REL_KEY]).abs().max()) + 1e-5


@pytest.mark.parametrize("model,loss", [("transe", "kl"), ("rotate", "kl"), ("complex", "bce")])
def test_sharded_negative_sampling_job_on_the_engine(model, loss):
    """ShardedTrainingJobNegativeSampling on the HIP engine (one rank: the shard + slack rows, kge_score_neg and
    kge_score_neg_bwd_accum on local ids): three SGD steps against TrainingJobNegativeSampling's step in float32 torch
    ops (train_negative_sampling.py:120-163).  float32 tables: losses to 1e-5 relative, parameters to 1e-4 of the
    step."""
    import torch.nn.functional as F
    import torch_port as tp
    from kge_amd.sharded_train import ENT_KEY, REL_KEY, ShardedTrainingJobNegativeSampling
    E, R, d, n, K, lr = 2003, 5, 128, 64, 50, 0.5
    dr = d // 2 if model == "rotate" else d
    g = torch.Generator().manual_seed(9)
    job = ShardedTrainingJobNegativeSampling(model, E, R, d, rel_dim=dr, seed=3, lr=lr, optimizer="SGD", device=DEV,
                                             n_max=n, loss=loss)
    sd0 = job.state_dict()
    ent = sd0[ENT_KEY].clone().requires_grad_(True)
    rel = sd0[REL_KEY].clone().requires_grad_(True)
    opt = torch.optim.SGD([ent, rel], lr=lr)
    for k in range(3):
        tri = torch.stack([torch.randint(hi, (n,), generator=g) for hi in (E, R, E)], 1)
        ns, no = torch.randint(E, (n, K), generator=g), torch.randint(E, (n, K // 2), generator=g)
        got = float(job.step(tri, ns, no))
        s, p, o = tri[:, 0], tri[:, 1], tri[:, 2]
        opt.zero_grad()
        want = 0.0
        for slot, neg in ((0, ns), (2, no)):
            kk = neg.shape[1]
            rep = lambda x: x.view(-1, 1).expand(n, kk).reshape(-1)
            pos = tp.score_emb(model, ent[s], rel[p], ent[o], "spo").view(-1)
            ss, oo = (neg.reshape(-1), rep(o)) if slot == 0 else (rep(s), neg.reshape(-1))
            block = torch.cat([pos.view(-1, 1), tp.score_emb(model, ent[ss], rel[rep(p)], ent[oo], "spo").view(n, kk)], 1)
            if loss == "kl":
                l = F.cross_entropy(block, torch.zeros(n, dtype=torch.long), reduction="sum") / n
            else:
                lab = torch.zeros_like(block)
                lab[:, 0] = 1.0
                l = F.binary_cross_entropy_with_logits(block, lab, reduction="sum") / n
            l.backward()
            want += float(l.detach())
        opt.step()
        assert abs(got - want) <= 2e-5 * abs(want) + 1e-6, (k, got, want)
    sd = job.state_dict()
    moved = float((ent.detach() - sd0[ENT_KEY]).abs().max())
    assert moved > 1e-5
    d_ent = (sd[ENT_KEY] - ent.detach()).abs()
    d_rel = (sd[REL_KEY] - rel.detach()).abs()
    b_ent = 1e-3 * moved + 1e-6
    b_rel = 1e-3 * float((rel.detach() - sd0[REL_KEY]).abs().max()) + 1e-6
    if model == "transe":
        # l_norm 1: d|q - t| / dq = sign(q - t).  A coordinate whose difference is rounding noise around zero after a
        # step (the float atomics of the gradient scatter order differently run by run) takes the other sign in the
        # next step and moves by 2 lr w / n: a handful of coordinates may, the rest must agree as everywhere else
        assert float((d_ent > b_ent).float().mean()) <= 5e-5 and float(d_ent.max()) <= 2 * lr / n, float(d_ent.max())
        assert float((d_rel > b_rel).float().mean()) <= 5e-3 and float(d_rel.max()) <= 2 * lr / n * 8, float(d_rel.max())
    else:
        assert float(d_ent.max()) <= b_ent
        assert float(d_rel.max()) <= b_rel
