#!/usr/bin/env python3
"""kge_score_sp / _po / _sp_po with MANY rows (one call, n >= 1024, d = 512, all entities): the route through the
persistent kernel (api.hip one_call_v8: ONE query-build launch + ONE pairs_bf16_v8_kernel launch over n / 512 batches)
against the route switched off (KGE_ONE_CALL_V8=0: a query build + the single-batch kernel on all n rows), both query
modes, FB15k-237 shape; through engine.score_sp / score_sp_po = what KgeModel.score_sp executes.  us per call
(HIP events around 30 calls, best of 3), fraction of 8 TB/s on the algorithmic bytes of SURVEY 8(d)."""
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
for split in (0, 1):
    T = engine.Tables("complex", ent, rel, flags=engine.FLAG_SPLIT_QUERY if split else 0)
    for n in (512, 1024, 2048, 4096, 8192):
        s, p, o = (torch.randint(hi, (n,), generator=g).to(dev) for hi in (E, R, E))
        row = {"split": split, "n": n}
        ref = {}
        for env in ("0", "1"):
            os.environ["KGE_ONE_CALL_V8"] = env
            for name, call in (("sp_po", lambda: engine.score_sp_po(T, s, p, o)), ("sp", lambda: engine.score_sp(T, s, p))):
                a = call()
                torch.cuda.synchronize()
                us = v8_probe.timed(call, 30, 3)
                row[f"{name}_v8_{env}_us"] = round(us, 2)
                ab = v8_probe.alg_bytes(n, E, D, 2 if name == "sp_po" else 1)
                row[f"{name}_v8_{env}_frac"] = round(ab / (us * 1e-6) / 8e12, 3)
                if env == "0":
                    ref[name] = a
                else:
                    row[f"{name}_bit_equal"] = bool(torch.equal(a, ref[name]))
            del a
        ref.clear()
        print(json.dumps(row), flush=True)
