#!/usr/bin/env python3
"""Forward / forward+backward of score_spo on n*K corrupted triples (negative sampling, `triple`
implementation, sampler.py:291-306), W shape, f32: ours vs the reference's torch ops on the GPU."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import model as km


def ref_spo(name, s_emb, p_emb, o_emb):
    """the reference's spo op sequences (complex.py:24-35, transe.py:18-22, rotate.py:30-41,146-213)"""
    if name == "transe":
        return -torch.nn.functional.pairwise_distance(s_emb + p_emb, o_emb, p=1.0)
    if name == "complex":
        p_re, p_im = (t.contiguous() for t in p_emb.chunk(2, dim=1))
        o_re, o_im = (t.contiguous() for t in o_emb.chunk(2, dim=1))
        s_all = torch.cat((s_emb, s_emb), dim=1)
        r_all = torch.cat((p_re, p_emb, -p_im), dim=1)
        o_all = torch.cat((o_emb, o_im, o_re), dim=1)
        return (s_all * o_all * r_all).sum(dim=1)
    s_re, s_im = s_emb.chunk(2, dim=1)
    o_re, o_im = o_emb.chunk(2, dim=1)
    cs, sn = torch.cos(p_emb), torch.sin(p_emb)
    d_re, d_im = s_re * cs - s_im * sn - o_re, s_re * sn + s_im * cs - o_im
    return -torch.stack((d_re, d_im), dim=0).norm(dim=0).sum(dim=1)


dev = torch.device("cuda", 0)
E, R, d = 40943, 11, 512
g = torch.Generator().manual_seed(0)


def timeit(fn, k=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / k


for name in ("rotate", "transe", "complex"):
    dr = d // 2 if name == "rotate" else d
    ent = torch.empty(E, d).normal_(0, 0.1, generator=g).to(dev).requires_grad_(True)
    rel = torch.empty(R, dr).normal_(0, 0.1, generator=g).to(dev).requires_grad_(True)
    n, K = 128, 1000
    s = torch.randint(E, (n,), generator=g).to(dev).repeat_interleave(K)
    p = torch.randint(R, (n,), generator=g).to(dev).repeat_interleave(K)
    o = torch.randint(E, (n * K,), generator=g).to(dev)
    w = torch.randn(n * K, device=dev)

    def ours_fwd():
        return km._ScoreSPO.apply(name, 1.0, ent, rel, s, p, o)

    def ours():
        ent.grad = rel.grad = None
        (ours_fwd() * w).sum().backward()

    def ref_fwd():
        return ref_spo(name, ent[s], rel[p], ent[o])

    def ref():
        ent.grad = rel.grad = None
        (ref_fwd() * w).sum().backward()

    with torch.no_grad():
        tf = timeit(ours_fwd)
    to = timeit(ours)
    try:
        with torch.no_grad():
            rf = timeit(ref_fwd)
        tr = timeit(ref)
    except Exception as e:
        rf = tr = float("nan"); print("torch ops failed:", type(e).__name__, str(e)[:200])
    print(f"{name:8s} N={n*K}: ours fwd {tf:8.1f} us, fwd+bwd {to:9.1f} us | torch ops fwd {rf:9.1f} us, fwd+bwd {tr:9.1f} us")
    # whole step: + Adagrad over both tables (dense gradients, as LibKGE's default sparse=False gives)
    from kge_amd.optim import Adagrad as HipAdagrad
    for tag, mk in (("torch.optim.Adagrad", lambda: torch.optim.Adagrad([ent, rel], lr=0.1)),
                    ("one-pass Adagrad", lambda: HipAdagrad([ent, rel], lr=0.1))):
        opt = mk()

        def step():
            ours()
            opt.step()
        print(f"           whole step with {tag:20s}: {timeit(step):9.1f} us")

# ---- kernel breakdown of the RotatE forward + backward
from torch.profiler import profile, ProfilerActivity
name = "rotate"
ent = torch.empty(E, d).normal_(0, 0.1, generator=g).to(dev).requires_grad_(True)
rel = torch.empty(R, d // 2).normal_(0, 0.1, generator=g).to(dev).requires_grad_(True)
n, K = 128, 1000
s = torch.randint(E, (n,), generator=g).to(dev).repeat_interleave(K)
p = torch.randint(R, (n,), generator=g).to(dev).repeat_interleave(K)
o = torch.randint(E, (n * K,), generator=g).to(dev)
w = torch.randn(n * K, device=dev)


def one():
    ent.grad = rel.grad = None
    (km._ScoreSPO.apply(name, 1.0, ent, rel, s, p, o) * w).sum().backward()


for _ in range(3): one()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5): one()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=10, max_name_column_width=60))
