#!/usr/bin/env python3
"""The split-query group launch (8 two-sided batches, FB15k-237 shape) under variants and store policies: us per launch
by HIP events over back-to-back launches on one stream.  Arguments = variants: "x4=0" / "x4=1" (KGE_V8_X4: dword stores /
16-byte stores of the split-query kernel), "var=1" (KGE_V8_VAR=1, a -DKGE_V8_PROBES build: store instructions left out)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kge_amd import engine  # noqa: E402
import v8_probe  # noqa: E402

dev = torch.device("cuda", 0)
E, R, D, n, L = 14541, 237, 512, 512, 8
P = engine.score_pitch(E)
g = torch.Generator().manual_seed(0)
ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
variants = sys.argv[1:] or ["x4=0", "x4=1"]
for split in (1, 0):
    fl = engine.FLAG_SPLIT_QUERY if split else None
    T = engine.Tables("complex", ent, rel, flags=fl or 0)
    groups = [torch.stack([torch.randint(hi, (n * L,), generator=g) for hi in (E, R, E)], 1).to(dev) for _ in range(2)]
    qs = [engine.QueriesGroup(T, "sp_po", n, L, flags=fl) for _ in range(2)]
    engine.build_queries_group(T, "sp_po", groups[0], n, L, out=qs[0])
    gbuf = torch.empty(L, n, 2 * P, device=dev)
    gout = gbuf.view(L, n, 2, P)[:, :, :, :E]
    kk = [0]

    def gstep():
        c = kk[0] & 1
        kk[0] += 1
        engine.score_queries_group(T, qs[c], gout, next_batch=groups[1 - c], next_queries=qs[1 - c])
    for rep in range(2):
        for var in variants:
            for pol in (None, "0", "1", "2"):
                key, val = var.split("=")
                os.environ.pop("KGE_V8_VAR", None)
                os.environ.pop("KGE_V8_X4", None)
                os.environ["KGE_V8_" + key.upper()] = val
                if pol is None:
                    os.environ.pop("KGE_V4_STORE_SC1", None)
                else:
                    os.environ["KGE_V4_STORE_SC1"] = pol
                us = v8_probe.timed(gstep, 40, 3)
                print(json.dumps({"split": split, "var": var, "policy": pol or "default", "rep": rep, "us_per_launch": round(us, 1),
                                  "us_per_batch": round(us / L, 2)}), flush=True)
    os.environ.pop("KGE_V8_VAR", None)
    os.environ.pop("KGE_V8_X4", None)
    os.environ.pop("KGE_V4_STORE_SC1", None)
