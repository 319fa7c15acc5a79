#!/usr/bin/env python3
"""pairs_bf16_v8_kernel (persistent, two consumer waves per SIMD; KGE_V8 unset) against the round-3 kernels
(KGE_V8=0: pairs_bf16_v7 / v6) at the FB15k-237 shape, prepared queries, rows on the 256-byte pitch:
  * one launch per batch (the pipelined step: every launch also builds the next batch's queries), one- and two-sided,
    plain and split queries;
  * groups of L batches in one launch (kge_score_queries_multi; every launch also builds the next group's queries).
HIP events around back-to-back launches, median of R rounds, variants alternating.  Then v8's cycle stamps.

    python tools/v8_probe.py [--steps 200] [--rounds 5]
"""
import argparse
import ctypes
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import _lib, engine  # noqa: E402

dev = torch.device("cuda", 0)
E, R, D = 14541, 237, 512
P = engine.score_pitch(E)


def alg_bytes(n, m, d, sides):
    return m * d * 2 + sides * (n * 2 * d * 2 + n * m * 4 + 2 * n * 8)


def timed(fn, steps, rounds):
    ts = []
    for _ in range(rounds):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e3)
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    n = 512
    for split in (0, 1):
        fl = engine.FLAG_SPLIT_QUERY if split else None
        T = engine.Tables("complex", ent, rel, flags=fl or 0)
        for comb, sides in (("sp_", 1), ("sp_po", 2)):
            batches = [torch.stack([torch.randint(hi, (n,), generator=g) for hi in (E, R, E)], 1).to(dev) for _ in range(2)]
            pipe = engine.ScorePipeline(T, comb, n, flags=fl)
            pipe.start(batches[0])
            buf = torch.empty(n, sides * P, device=dev)
            out = buf.view(n, 2, P)[:, :, :E] if sides == 2 else buf[:, :E]
            k = [0]

            def step():
                k[0] += 1
                pipe.step(next_batch=batches[k[0] & 1], out=out)
            row = {"launch": "one batch", "n": n, "combine": comb, "split": split}
            for name, env in (("v8", None), ("r3", "0")):
                if env is None:
                    os.environ.pop("KGE_V8", None)
                else:
                    os.environ["KGE_V8"] = env
                us = timed(step, a.steps, a.rounds)
                row[name + "_us"] = round(us, 2)
                row[name + "_frac"] = round(alg_bytes(n, E, D, sides) / (us * 1e-6) / 8e12, 3)
            os.environ.pop("KGE_V8", None)
            print(json.dumps(row), flush=True)
            del pipe
            # groups
            for L in (2, 4, 8, 16):
                groups = [torch.stack([torch.randint(hi, (n * L,), generator=g) for hi in (E, R, E)], 1).to(dev)
                          for _ in range(2)]
                qs = [engine.QueriesGroup(T, comb, n, L, flags=fl) for _ in range(2)]
                engine.build_queries_group(T, comb, groups[0], n, L, out=qs[0])
                gbuf = torch.empty(L, n, sides * P, device=dev)
                gout = gbuf.view(L, n, 2, P)[:, :, :, :E] if sides == 2 else gbuf[:, :, :E]
                kk = [0]

                def gstep():
                    c = kk[0] & 1
                    kk[0] += 1
                    engine.score_queries_group(T, qs[c], gout, next_batch=groups[1 - c], next_queries=qs[1 - c])
                us = timed(gstep, max(10, a.steps // L), a.rounds)
                print(json.dumps({"launch": f"group of {L}", "n": n, "combine": comb, "split": split,
                                  "us_per_launch": round(us, 2), "us_per_batch": round(us / L, 2),
                                  "frac": round(L * alg_bytes(n, E, D, sides) / (us * 1e-6) / 8e12, 3),
                                  "frac_table_once": round((alg_bytes(n, E, D, sides) * L - (L - 1) * E * D * 2)
                                                           / (us * 1e-6) / 8e12, 3),
                                  "score_MB": round(L * n * sides * P * 4 / 1e6)}), flush=True)
                del gbuf, gout, qs
                torch.cuda.empty_cache()
    # ---- stamps of the v8 launch
    L_ = _lib.lib()
    L_.kge_debug_v6_stamps.restype = None
    L_.kge_debug_v6_stamps.argtypes = [ctypes.c_void_p]
    T = engine.Tables("complex", ent, rel)
    for comb, sides, L in (("sp_", 1, 1), ("sp_po", 2, 1), ("sp_po", 2, 8)):
        grp = torch.stack([torch.randint(hi, (n * L,), generator=g) for hi in (E, R, E)], 1).to(dev)
        q = engine.build_queries_group(T, comb, grp, n, L)
        gbuf = torch.empty(L, n, sides * P, device=dev)
        gout = gbuf.view(L, n, 2, P)[:, :, :, :E] if sides == 2 else gbuf[:, :, :E]
        for _ in range(3):
            engine.score_queries_group(T, q, gout)
        st = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        L_.kge_debug_v6_stamps(ctypes.c_void_p(st.data_ptr()))
        engine.score_queries_group(T, q, gout)
        torch.cuda.synchronize()
        L_.kge_debug_v6_stamps(None)
        v = st.view(4096, 64).cpu()
        v = v[(v[:, 0] != 0) & (v[:, 2] != 0)]
        nst = int((v[0, :32] != 0).sum())
        rel_ = (v[:, :nst] - v[:, :1]).double().median(dim=0).values
        per = [float(rel_[i + 1] - rel_[i]) for i in range(2, nst - 1)]
        last = (v[:, 34] - v[:, 0]).double()
        print(f"==== v8 stamps {comb} group of {L}: {v.shape[0]} workgroups; R0 passed {float(rel_[1]):.0f}, first chain issued "
              f"{float(rel_[2]):.0f}; unit periods {[round(x) for x in per[:12]]}; median {statistics.median(per) if per else 0:.0f}; "
              f"last store issued median {float(last.median()):.0f} max {float(last.max()):.0f}; "
              f"starts spread {int(v[:, 0].max() - v[:, 0].min())}")


if __name__ == "__main__":
    main()
