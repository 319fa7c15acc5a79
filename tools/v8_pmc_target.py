#!/usr/bin/env python3
"""The workload of the round-4 HBM counter passes: 12 group launches (8 two-sided batches of 512, FB15k-237 shape,
ComplEx d = 512, prepared queries, every launch also builds the next group's queries) in each query mode -- what
bench.py's timed region issues.  tools/gpu_r4prof.sh runs it under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
E, R, D, N, L = 14541, 237, 512, 512, int(os.environ.get("GROUP", "8"))
P = engine.score_pitch(E)
g = torch.Generator().manual_seed(0)
ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
for flags in (engine.FLAG_SPLIT_QUERY, 0):
    T = engine.Tables("complex", ent, rel, flags=flags)
    tri = [torch.stack([torch.randint(hi, (N * L,), generator=g) for hi in (E, R, E)], 1).to(dev) for _ in range(2)]
    qs = [engine.QueriesGroup(T, "sp_po", N, L, flags=flags) for _ in range(2)]
    engine.build_queries_group(T, "sp_po", tri[0], N, L, out=qs[0])
    buf = torch.empty(L, N, 2 * P, device=dev)
    out = buf.view(L, N, 2, P)[:, :, :, :E]
    for k in range(12):
        c = k & 1
        engine.score_queries_group(T, qs[c], out, next_batch=tri[1 - c], next_queries=qs[1 - c])
    torch.cuda.synchronize()
