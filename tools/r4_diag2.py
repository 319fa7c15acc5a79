#!/usr/bin/env python3
"""Round-4 diagnosis, second pass: what sets the 1.44 k-cycle unit period of pairs_bf16_v7_kernel when its stores cost
nothing (tools/r4_diag.py: the period is the same with every store dropped).  KGE_V7_PROBE = compile-time variants of
the kernel, bits: 1 no store instructions, 2 no table DMA behind the ring fill (stale units), 4 no fragment reads from
LDS in the chains (stale registers), 8 no workgroup barrier per unit.  Timing only (wrong scores); the variants are
in the library only when it is built with `make -C kge_amd/csrc -B CXXEXTRA=-DKGE_V7_PROBES`."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kge_amd import engine  # noqa: E402
import r4_diag  # noqa: E402

dev = torch.device("cuda", 0)
E, R, D = r4_diag.E, r4_diag.R, r4_diag.D
P = r4_diag.P


def main():
    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    T = engine.Tables("complex", ent, rel)
    for n in (512, 4096):
        batches = [tuple(torch.randint(hi, (n,), generator=g).to(dev) for hi in (E, R, E)) for _ in range(2)]
        pipe = engine.ScorePipeline(T, "sp_", n)
        pipe.start(*batches[0])
        buf = torch.empty(n, P, device=dev)
        out = buf[:, :E]
        k = [0]

        def step():
            k[0] += 1
            pipe.step(next_batch=batches[k[0] & 1], out=out)
        for sc1 in ("0",):
            for probe in (0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 0):
                os.environ["KGE_V4_STORE_SC1"] = sc1
                os.environ["KGE_V7_PROBE"] = str(probe)
                us = r4_diag.timed(step, max(20, 300 * 512 // n))
                s = r4_diag.stamps(step)
                print(json.dumps({"n": n, "sc1": sc1, "probe": probe, "us": round(us, 2),
                                  "unit_period_median": s["unit_period_median"], "R0": s["R0"],
                                  "first_chain_issued": s["first_chain_issued"],
                                  "last_store_issued": s["last_store_issued"],
                                  "periods": s["unit_period_first8"]}), flush=True)
        os.environ.pop("KGE_V7_PROBE", None)
        os.environ.pop("KGE_V4_STORE_SC1", None)


if __name__ == "__main__":
    main()
