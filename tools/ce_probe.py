#!/usr/bin/env python3
"""Fused 1vsAll loss (kge_ce_fwd / kge_ce_bwd) against the unfused mixed-precision path at the
BASELINE configs[1] shape (E=14541, d=512, n=512, bf16 tables): kernel-level times from the torch
profiler and whole forward / forward+backward times from HIP events."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
E, R, d, n = 14541, 237, 512, 512
g = torch.Generator().manual_seed(0)
ent = torch.empty(E, d).normal_(0, 0.1, generator=g).to(torch.bfloat16).to(dev)
rel = torch.empty(R, d).normal_(0, 0.1, generator=g).to(torch.bfloat16).to(dev)
s = torch.randint(E, (n,), generator=g).to(dev)
p = torch.randint(R, (n,), generator=g).to(dev)
o = torch.randint(E, (n,), generator=g).to(dev)


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for model in ("complex", "distmult"):
    T = engine.Tables(model, ent, rel)

    def fused_fwd():
        return engine.ce_fwd(T, "sp", s, p, o)

    def fused_fwd_bwd():
        loss, lse = engine.ce_fwd(T, "sp", s, p, o)
        return engine.ce_bwd(T, "sp", s, p, o, lse, g_scalar=1.0 / n)

    def unfused_fwd():
        sc = engine.score_sp(T, s, p)
        return torch.nn.functional.cross_entropy(sc, o, reduction="sum")

    def unfused_fwd_bwd():
        sc = engine.score_sp(T, s, p)
        sc.requires_grad_(True)
        loss = torch.nn.functional.cross_entropy(sc, o, reduction="sum") / n
        (ds,) = torch.autograd.grad(loss, sc)
        return engine.score_pairs_bwd(T, "sp", s, p, None, ds)

    print(f"{model}: forward  fused {timeit(fused_fwd):7.1f} us | unfused (score_sp + F.cross_entropy) {timeit(unfused_fwd):7.1f} us")
    print(f"{model}: fwd+bwd  fused {timeit(fused_fwd_bwd):7.1f} us | unfused (+ autograd softmax grad + kge_score_pairs_bwd) {timeit(unfused_fwd_bwd):7.1f} us")
    if model == "complex":
        from torch.profiler import ProfilerActivity, profile
        for nm, fn in (("fused", fused_fwd_bwd), ("unfused", unfused_fwd_bwd)):
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
            print(f"--- {nm} fwd+bwd, 10 iterations, kernels by total time")
            rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:12]
            for e in rows:
                print(f"  {e.device_time_total / 10:8.1f} us/iter  x{e.count / 10:4.1f}  {e.key[:90]}")
