#!/usr/bin/env python3
"""The float32 backward of ComplEx sp_ scores at the FB15k-237 shape (engine.score_pairs_bwd: query build, the two
gemm32_kernel products, chain rule) for a kernel trace:  rocprofv3 --kernel-trace --stats -- python tools/bwd32_probe.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine  # noqa: E402
dev = torch.device("cuda", 0)
E, R, D, n = 14541, 237, 512, 512
g = torch.Generator().manual_seed(0)
ent = torch.empty(E, D).normal_(0, 0.1, generator=g).to(dev)
rel = torch.empty(R, D).normal_(0, 0.1, generator=g).to(dev)
s, p = torch.randint(E, (n,), generator=g).to(dev), torch.randint(R, (n,), generator=g).to(dev)
T = engine.Tables("complex", ent, rel)
scores = engine.score_sp(T, s, p)
gout = torch.randn_like(scores)
for _ in range(30):
    engine.score_pairs_bwd(T, "sp", s, p, None, gout, scores)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    engine.score_pairs_bwd(T, "sp", s, p, None, gout, scores)
e1.record(); torch.cuda.synchronize()
print("score_pairs_bwd f32 complex n=512: %.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
q = torch.randn(n, D, device=dev)
for name, fn in (("torch.mm gout @ ent", lambda: gout @ ent), ("torch.mm gout.t() @ q", lambda: gout.t() @ q)):
    for _ in range(5): fn()
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print("%s: %.1f us" % (name, e0.elapsed_time(e1) / 50 * 1e3))
