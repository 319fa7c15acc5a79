#!/usr/bin/env python3
"""Workload for the rocprofv3 passes on the gather-bound kernels: kge_score_neg (TransE, RotatE,
ComplEx; f32 tables) and kge_score_spo at the WN18RR negative-sampling shape
(E=40943, d=512, 512 positives x 1000 negatives).  Run under
  rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE   (separate passes)
by tools/gpu_pmc.sh; tools/pmc_summary.py --neg turns the three into HBM GB/s per kernel."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
E, R, d, n, K = 40943, 11, 512, 512, 1000
g = torch.Generator().manual_seed(3)
s = torch.randint(E, (n,), generator=g).to(dev)
p = torch.randint(R, (n,), generator=g).to(dev)
o = torch.randint(E, (n,), generator=g).to(dev)
neg = torch.randint(E, (n, K), generator=g).to(dev)
for model in ("transe", "rotate", "complex"):
    ent = torch.empty(E, d).normal_(0, 0.1, generator=g).to(dev)
    dr = d // 2 if model == "rotate" else d
    rel = torch.empty(R, dr).uniform_(-3.14, 3.14, generator=g).to(dev)
    T = engine.Tables(model, ent, rel)
    for _ in range(int(os.environ.get("NEG_PMC_ITERS", "12"))):
        engine.score_neg(T, s, p, o, 2, neg)
        engine.score_neg(T, s, p, o, 0, neg)
    torch.cuda.synchronize()
print("ok")
