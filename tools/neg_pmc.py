#!/usr/bin/env python3
"""Workload for the rocprofv3 passes on the gather-bound kernel kge_score_neg (BASELINE configs[2]: 512 positives x
1,000 negatives per slot, d = 512, float32; RotatE and TransE) -- the launches bench.py's `roofline_neg` leg times:
NEG_PMC_CASE=wn18rr (E = 40,943: the table fits the Infinity Cache) or =big (E = 2,000,000: it does not).  Run under
  rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE   (separate passes, tools/gpu_r3neg.sh);
tools/neg_pmc_summary.py turns them into profiles/<tag>_rocprofv3_neg.txt and profiles/pmc_neg_latest.json."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
case = os.environ.get("NEG_PMC_CASE", "wn18rr")
E = 40943 if case == "wn18rr" else 2000000
R, d, n, K = 11, 512, 512, 1000
g = torch.Generator(device=dev).manual_seed(5)
ent = torch.empty(E, d, device=dev).normal_(0, 0.1, generator=g)
q = torch.Generator().manual_seed(6)
s, o = (torch.randint(E, (n,), generator=q).to(dev) for _ in range(2))
p = torch.randint(R, (n,), generator=q).to(dev)
neg = torch.randint(E, (n, K), generator=q).to(dev)
for model in ("rotate", "transe"):
    dr = d // 2 if model == "rotate" else d
    rel = torch.empty(R, dr, device=dev).uniform_(-3.14, 3.14, generator=g)
    T = engine.Tables(model, ent, rel)
    for _ in range(int(os.environ.get("NEG_PMC_ITERS", "12"))):
        engine.score_neg(T, s, p, o, 2, neg)
    torch.cuda.synchronize()
print("ok")
