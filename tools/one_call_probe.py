#!/usr/bin/env python3
"""One-call entry points (kge_score_sp / kge_score_sp_po) at the FB15k-237 shape: builder launch + prepared scoring
launch (KGE_ONE_CALL_PREPARED=1) against the cooperative in-launch build (=0)."""
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
E, R, D = 14541, 237, 512
g = torch.Generator().manual_seed(0)
ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
T = engine.Tables("complex", ent, rel)
for n in (64, 128, 512, 1024):
    s, p, o = (torch.randint(hi, (n,), generator=g).to(dev) for hi in (E, R, E))
    row = {"n": n}
    for env in ("0", "1"):
        os.environ["KGE_ONE_CALL_PREPARED"] = env
        a = engine.score_sp_po(T, s, p, o)
        row[f"sp_po_prepared{env}_us"] = round(v8_probe.timed(lambda: engine.score_sp_po(T, s, p, o), 200, 3), 2)
        row[f"sp_prepared{env}_us"] = round(v8_probe.timed(lambda: engine.score_sp(T, s, p), 200, 3), 2)
        if env == "0":
            ref = a
        else:
            row["bit_equal"] = bool(torch.equal(a, ref))
    ab2, ab1 = v8_probe.alg_bytes(n, E, D, 2), v8_probe.alg_bytes(n, E, D, 1)
    row["sp_po_frac_prepared"] = round(ab2 / (row["sp_po_prepared1_us"] * 1e-6) / 8e12, 3)
    row["sp_frac_prepared"] = round(ab1 / (row["sp_prepared1_us"] * 1e-6) / 8e12, 3)
    print(json.dumps(row), flush=True)
