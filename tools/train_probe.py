#!/usr/bin/env python3
"""One 1vsAll training step (train_1vsAll.py:48-82: score_sp -> KL/CE loss -> backward, score_po ->
loss -> backward, Adagrad step) at the C2 shape with float32 parameters: the stand-alone mirror
model (f32 scoring, mixed-precision scoring) vs the reference's op sequence in PyTorch-ROCm."""
import os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import model as km

dev = torch.device("cuda", 0)
E, R, d, n = 14541, 237, 512, 512
g = torch.Generator().manual_seed(0)
s = torch.randint(E, (n,), generator=g).to(dev); p = torch.randint(R, (n,), generator=g).to(dev)
o = torch.randint(E, (n,), generator=g).to(dev)


class RefModel(torch.nn.Module):  # complex.py:24-39 / distmult.py:15-21 on LookupEmbedder tables
    def __init__(self, name):
        super().__init__()
        self.name = name
        self.ent = torch.nn.Parameter(torch.empty(E, d, device=dev).normal_(0, 0.1))
        self.rel = torch.nn.Parameter(torch.empty(R, d, device=dev).normal_(0, 0.1))
        self.all = torch.arange(E, device=dev)

    def _emb(self, s_emb, p_emb, o_emb, combine):
        if self.name == "distmult":
            return (s_emb * p_emb).mm(o_emb.t()) if combine == "sp_" else (o_emb * p_emb).mm(s_emb.t())
        p_re, p_im = (t.contiguous() for t in p_emb.chunk(2, dim=1))
        o_re, o_im = (t.contiguous() for t in o_emb.chunk(2, dim=1))
        s_all = torch.cat((s_emb, s_emb), dim=1)
        r_all = torch.cat((p_re, p_emb, -p_im), dim=1)
        o_all = torch.cat((o_emb, o_im, o_re), dim=1)
        return (s_all * r_all).mm(o_all.t()) if combine == "sp_" else (r_all * o_all).mm(s_all.t())

    def score_sp(self, s, p):
        return self._emb(self.ent[s], self.rel[p], self.ent[self.all], "sp_")

    def score_po(self, p, o):
        return self._emb(self.ent[self.all], self.rel[p], self.ent[o], "_po")


def step(m, opt):
    opt.zero_grad(set_to_none=True)
    F.cross_entropy(m.score_sp(s, p), o, reduction="sum").backward()
    F.cross_entropy(m.score_po(p, o), s, reduction="sum").backward()
    opt.step()


def step_fused2(m, opt):  # both directions from one scoring launch / one pair of gradient products
    opt.zero_grad(set_to_none=True)
    m.loss_sp_po(s, p, o).sum().backward()
    opt.step()


def step_fused(m, opt):  # loss fused into the scoring kernel (kge_ce_fwd / kge_ce_bwd)
    opt.zero_grad(set_to_none=True)
    m.loss_sp(s, p, o).sum().backward()
    m.loss_po(p, o, s).sum().backward()
    opt.step()


def timeit(fn, k=20, reps=5):
    """best of `reps` runs of k steps (the step is close to host-bound: the host's state matters)"""
    for _ in range(5): fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): fn()
        torch.cuda.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0) / k)
    return best


for name in ("complex", "distmult"):
    res = {}
    for tag, mk in (("reference ops (PyTorch-ROCm)", lambda: RefModel(name)),
                    ("kge_amd f32 scoring", lambda: km.create(name, E, R, d, device=dev)),
                    ("kge_amd score_dtype=bfloat16", lambda: km.create(name, E, R, d, device=dev, score_dtype=torch.bfloat16))):
        m = mk()
        opt = torch.optim.Adagrad(m.parameters(), lr=0.1)
        res[tag] = timeit(lambda: step(m, opt))
    m = km.create(name, E, R, d, device=dev, score_dtype=torch.bfloat16)
    opt = torch.optim.Adagrad(m.parameters(), lr=0.1)
    res["kge_amd score_dtype=bfloat16, fused loss"] = timeit(lambda: step_fused(m, opt))
    from kge_amd.optim import Adagrad as HipAdagrad
    m = km.create(name, E, R, d, device=dev, score_dtype=torch.bfloat16)
    opt = HipAdagrad(m.parameters(), lr=0.1, bf16_copies=True)
    res["... + one-pass Adagrad with bf16 copies"] = timeit(lambda: step_fused(m, opt))
    m = km.create(name, E, R, d, device=dev, score_dtype=torch.bfloat16)
    opt = HipAdagrad(m.parameters(), lr=0.1, bf16_copies=True)
    res["... + both directions in one pass"] = timeit(lambda: step_fused2(m, opt))
    # the same step captured once into a hipGraph and replayed: no per-launch host work at all
    m = km.create(name, E, R, d, device=dev, score_dtype=torch.bfloat16)
    opt = HipAdagrad(m.parameters(), lr=0.1, bf16_copies=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step_fused2(m, opt)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        m.loss_sp_po(s, p, o).sum().backward()
        opt.step()
    res["... replayed as one hipGraph"] = timeit(graph.replay)
    print(name, " | ".join(f"{k}: {v:.2f} ms" for k, v in res.items()))

# ---- where the mixed-precision step spends its GPU time
from torch.profiler import profile, ProfilerActivity
m = km.create("complex", E, R, d, device=dev, score_dtype=torch.bfloat16)
opt = torch.optim.Adagrad(m.parameters(), lr=0.1)
from kge_amd.optim import Adagrad as HipAdagrad
m2 = km.create("complex", E, R, d, device=dev, score_dtype=torch.bfloat16)
opt2 = HipAdagrad(m2.parameters(), lr=0.1, bf16_copies=True)
for tag, fn, m, opt in (("composed loss", step, m, opt), ("fused loss", step_fused, m, opt),
                        ("fused loss, both directions in one pass + one-pass Adagrad with bf16 copies", step_fused2, m2, opt2)):
    for _ in range(3): fn(m, opt)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5): fn(m, opt)
        torch.cuda.synchronize()
    print(f"---- mixed precision, {tag}: 5 steps")
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
