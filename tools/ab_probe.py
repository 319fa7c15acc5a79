#!/usr/bin/env python3
"""A/B of launch-level switches of the prepared-query scoring launch at the FB15k-237 shape: every variant is timed
R times in turn (ABCABC...: clocks and box noise hit all variants alike), S back-to-back steps each; median and
minimum per variant.   python tools/ab_probe.py [--steps 400] [--rounds 7]"""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--n", type=int, default=512)
    a = ap.parse_args()
    n = a.n
    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    T = engine.Tables("complex", ent, rel)
    batches = [tuple(torch.randint(hi, (n,), generator=g).to(dev) for hi in (E, R, E)) for _ in range(2)]
    variants = []
    for comb in ("sp_", "sp_po"):
        w = E if comb == "sp_" else 2 * E
        for pad in (0, 1):
            ld = (w + 63) // 64 * 64 if pad else w
            buf = torch.empty(n, ld, device=dev)
            out = buf[:, :w]
            for sc1 in ("0", "1"):
                for il in ("0", "1"):
                    variants.append((comb, pad, sc1, il, out))
    pipes = {c: engine.ScorePipeline(T, c, n) for c in ("sp_", "sp_po")}
    for c, pp in pipes.items():
        pp.start(*batches[0])
    res = {i: [] for i in range(len(variants))}
    k = [0]
    for r in range(a.rounds):
        for i, (comb, pad, sc1, il, out) in enumerate(variants):
            os.environ["KGE_V4_STORE_SC1"] = sc1
            os.environ["KGE_V4_INTERLEAVE"] = il
            pp = pipes[comb]

            def step():
                k[0] += 1
                pp.step(next_batch=batches[k[0] & 1], out=out)
            for _ in range(20):
                step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.steps):
                step()
            e1.record()
            torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) / a.steps * 1e3)
    for i, (comb, pad, sc1, il, out) in enumerate(variants):
        v = res[i]
        print(json.dumps({"combine": comb, "padded_pitch": pad, "store_sc1": sc1, "interleave": il,
                          "median_us": round(statistics.median(v), 2), "min_us": round(min(v), 2),
                          "all": [round(x, 2) for x in v]}))


if __name__ == "__main__":
    main()
