#!/usr/bin/env python3
"""Round-4 diagnosis of the direct-store scoring launch (pairs_bf16_v7_kernel), FB15k-237 shape, prepared queries:
what bounds it once start-up, tail and idle compute units are amortised -- i.e. what a multi-batch launch can reach.

  * one-sided launches of n = 512 .. 8192 rows (n = 4096 IS eight batches of 512 in one launch): back-to-back HIP-event
    time, per-unit cycle stamps of consumer wave 0, cycles / time = the clock;
  * the same with write-through / plain stores, with every store dropped (KGE_V7_NOSTORE=1: what the loads + matrix
    pipe alone take), with contiguous / interleaved unit ranges;
  * two-sided n = 512 into ONE score buffer (rewritten in the Infinity Cache) vs a rotation of buffers (1.9 GB: HBM).

    python tools/r4_diag.py > gpurun_out/<tag>/r4_diag.txt
"""
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


def timed(fn, steps, rounds=3):
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


def stamps(fn):
    """One launch of `fn` with the kernel's cycle stamps on: per workgroup, consumer wave 0's stamps (start,
    fragments requested, R0, then one per chain) and lane-side stamps in slots 32..."""
    L = _lib.lib()
    L.kge_debug_v6_stamps.restype = None
    L.kge_debug_v6_stamps.argtypes = [ctypes.c_void_p]
    st = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    L.kge_debug_v6_stamps(ctypes.c_void_p(st.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    L.kge_debug_v6_stamps(None)
    us = e0.elapsed_time(e1) * 1e3
    v = st.view(4096, 64).cpu()
    v = v[(v[:, 0] != 0) & (v[:, 3] != 0)]
    own = v[:, :32].clone()
    nst = int((own[0] != 0).sum())
    rel = (own[:, :nst] - own[:, :1]).double().median(dim=0).values
    chains = rel[3:]
    per = [float(chains[i + 1] - chains[i]) for i in range(len(chains) - 1)]
    span = int(own[:, :nst].max() - own[:, 0].min())
    last_store = (v[:, 34] - v[:, 0]).double().median().item() if bool((v[:, 34] != 0).any()) else None
    return {"workgroups": int(v.shape[0]), "R0": float(rel[2]), "first_chain_issued": float(rel[3]),
            "unit_period_median": statistics.median(per) if per else None,
            "unit_period_first8": [round(x) for x in per[:8]], "unit_period_last8": [round(x) for x in per[-8:]],
            "span_cycles": span, "launch_us_with_stamps": round(us, 2),
            "last_store_issued": last_store, "starts_spread": int(v[:, 0].max() - v[:, 0].min())}


def main():
    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    T = engine.Tables("complex", ent, rel)
    envs = {
        "sc1": {},
        "plain": {"KGE_V4_STORE_SC1": "0"},
        "nostore": {"KGE_V7_NOSTORE": "1"},
        "contig": {"KGE_V4_INTERLEAVE": "0"},
        "interleave": {"KGE_V4_INTERLEAVE": "1"},
    }
    keys = ("KGE_V4_STORE_SC1", "KGE_V7_NOSTORE", "KGE_V4_INTERLEAVE")
    for n in (512, 1024, 2048, 4096, 8192):
        batches = [tuple(torch.randint(hi, (n,), generator=g).to(dev) for hi in (E, R, E)) for _ in range(2)]
        pipe = engine.ScorePipeline(T, "sp_", n)
        pipe.start(*batches[0])
        buf = torch.empty(n, P, device=dev)
        out = buf[:, :E]
        k = [0]

        def step():
            k[0] += 1
            pipe.step(next_batch=batches[k[0] & 1], out=out)
        steps = max(20, min(300, 300 * 512 // n))
        for name, env in envs.items():
            if name in ("contig", "interleave") and n < 2048:
                continue
            for kk in keys:
                os.environ.pop(kk, None)
            os.environ.update(env)
            us = timed(step, steps)
            row = {"case": "one_sided", "n": n, "variant": name, "us": round(us, 2),
                   "frac": round(alg_bytes(n, E, D, 1) / (us * 1e-6) / 8e12, 3),
                   "write_TBps": round(n * E * 4 / us * 1e-6, 2)}
            s = stamps(step)
            row.update(s)
            if s["span_cycles"] and us:
                row["clock_GHz_est"] = round(s["span_cycles"] / (s["launch_us_with_stamps"] * 1e3), 2)
            print(json.dumps(row), flush=True)
        for kk in keys:
            os.environ.pop(kk, None)
        del buf, out, pipe
        torch.cuda.empty_cache()
    # ---- two-sided n = 512: one buffer (Infinity Cache) vs a rotation (HBM)
    n = 512
    batches = [tuple(torch.randint(hi, (n,), generator=g).to(dev) for hi in (E, R, E)) for _ in range(2)]
    for nbuf in (1, 2, 4, 8, 32):
        pipe = engine.ScorePipeline(T, "sp_po", n)
        pipe.start(*batches[0])
        bufs = [torch.empty(n, 2 * P, device=dev) for _ in range(nbuf)]
        outs = [b.view(n, 2, P)[:, :, :E] for b in bufs]
        k = [0]

        def step2():
            k[0] += 1
            pipe.step(next_batch=batches[k[0] & 1], out=outs[k[0] % nbuf])
        for name in ("sc1", "plain"):
            for kk in keys:
                os.environ.pop(kk, None)
            os.environ.update(envs[name])
            us = timed(step2, 256)
            print(json.dumps({"case": "two_sided_rotation", "n": n, "buffers": nbuf, "variant": name,
                              "MB_in_rotation": round(nbuf * n * 2 * P * 4 / 1e6), "us": round(us, 2),
                              "frac": round(alg_bytes(n, E, D, 2) / (us * 1e-6) / 8e12, 3)}), flush=True)
        for kk in keys:
            os.environ.pop(kk, None)
        del bufs, outs, pipe
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
