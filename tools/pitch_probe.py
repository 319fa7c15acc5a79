#!/usr/bin/env python3
"""Score-row pitch against the write stream of the prepared-query scoring launch (FB15k-237 shape): the same launch
with the rows of the score block on different pitches (floats).   python tools/pitch_probe.py"""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
E, R, D = 14541, 237, 512


def alg_bytes(n, m, d, sides):
    return m * d * 2 + sides * (n * 2 * d * 2 + n * m * 4 + 2 * n * 8)


def main():
    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    T = engine.Tables("complex", ent, rel)
    steps = 200
    for n, comb, sides in ((512, "sp_po", 2), (2048, "sp_", 1), (512, "sp_", 1)):
        batches = [tuple(torch.randint(hi, (n,), generator=g).to(dev) for hi in (E, R, E)) for _ in range(2)]
        pipe = engine.ScorePipeline(T, comb, n)
        pipe.start(*batches[0])
        res = {}
        for P in (14541, 14544, 14592, 14656, 14720, 14848, 15360, 16384, 14600, 14608, 14624):
            if sides == 2 and P == 14541:
                buf = torch.empty(n, 2 * E, device=dev)
                out = buf
            else:
                buf = torch.empty(n, sides * P, device=dev)
                out = buf.view(n, sides, P)[:, :, :E] if sides == 2 else buf[:, :E]
            k = [0]

            def step():
                k[0] += 1
                pipe.step(next_batch=batches[k[0] & 1], out=out)
            ts = []
            for r in range(3):
                for _ in range(20):
                    step()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(steps):
                    step()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / steps * 1e3)
            md = statistics.median(ts)
            res[P] = (round(md, 2), round(alg_bytes(n, E, D, sides) / (md * 1e-6) / 8e12, 3))
        print(json.dumps({"n": n, "combine": comb, "pitch_floats -> (us, frac)": res}), flush=True)


if __name__ == "__main__":
    main()
