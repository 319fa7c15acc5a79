#!/usr/bin/env python3
"""Cache policy of pairs_bf16_v8_kernel's score stores (KGE_V4_STORE_SC1 = 0 plain, 1 sc1, 2 nt, 3 sc1 nt) by group
size at the FB15k-237 shape: us per batch, HIP events around back-to-back group launches."""
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
E, R, D, P = v8_probe.E, v8_probe.R, v8_probe.D, v8_probe.P


def main():
    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    n = 512
    T = engine.Tables("complex", ent, rel)
    for comb, sides in (("sp_", 1), ("sp_po", 2)):
        for L in (1, 2, 4, 8, 16):
            groups = [torch.stack([torch.randint(hi, (n * L,), generator=g) for hi in (E, R, E)], 1).to(dev) for _ in range(2)]
            qs = [engine.QueriesGroup(T, comb, n, L) for _ in range(2)]
            engine.build_queries_group(T, comb, groups[0], n, L, out=qs[0])
            gbuf = torch.empty(L, n, sides * P, device=dev)
            gout = gbuf.view(L, n, 2, P)[:, :, :, :E] if sides == 2 else gbuf[:, :, :E]
            kk = [0]

            def gstep():
                c = kk[0] & 1
                kk[0] += 1
                engine.score_queries_group(T, qs[c], gout, next_batch=groups[1 - c], next_queries=qs[1 - c])
            row = {"combine": comb, "L": L, "score_MB": round(L * n * sides * P * 4 / 1e6)}
            for pol in ("0", "1", "2", "3"):
                os.environ["KGE_V4_STORE_SC1"] = pol
                us = v8_probe.timed(gstep, max(10, 200 // L), 3)
                row["us_per_batch_policy" + pol] = round(us / L, 2)
            os.environ.pop("KGE_V4_STORE_SC1", None)
            row["best_frac"] = round(v8_probe.alg_bytes(n, E, D, sides) /
                                     (min(v for k, v in row.items() if k.startswith("us_per")) * 1e-6) / 8e12, 3)
            print(json.dumps(row), flush=True)
            del gbuf, gout, qs
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
