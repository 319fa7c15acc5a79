#!/usr/bin/env python3
"""Prepared / split queries against the in-launch cooperative build at the FB15k-237 shape (BASELINE configs[1]):
HIP-event averages of back-to-back calls (one line of JSON per variant) + the phase stamps of the prepared launch.

    python tools/prep_probe.py [--steps 200]
"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import _lib, engine  # noqa: E402

dev = torch.device("cuda", 0)
E, R, D = 14541, 237, 512


def ev(fn, steps):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3  # us


def alg_bytes(n, m, d, sides):
    return m * d * 2 + sides * (n * 2 * d * 2 + n * m * 4 + 2 * n * 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    for scorer, sc1 in (("complex", "1"), ("complex", "0")):
        os.environ["KGE_V4_STORE_SC1"] = sc1
        print(f"---- KGE_V4_STORE_SC1={sc1}")
        T = engine.Tables(scorer, ent, rel)
        TS = engine.Tables(scorer, ent, rel, flags=engine.FLAG_SPLIT_QUERY)
        for n in ((512, 128, 1024, 2048) if sc1 == "0" else (512,)):
            q = torch.Generator().manual_seed(n)
            batches = [tuple(torch.randint(hi, (n,), generator=q).to(dev) for hi in (E, R, E)) for _ in range(2)]
            s, p, o = batches[0]
            out1 = torch.empty(n, E, device=dev)
            out2 = torch.empty(n, 2 * E, device=dev)
            res = {"scorer": scorer, "n": n}
            res["coop_one_sided_us"] = ev(lambda: engine.score_sp(T, s, p), a.steps)
            res["coop_two_sided_us"] = ev(lambda: engine.score_sp_po(T, s, p, o), a.steps)
            for comb, out, sides in (("sp_", out1, 1), ("sp_po", out2, 2)):
                for tag, TT, fl in (("", T, None), ("split_", TS, engine.FLAG_SPLIT_QUERY)):
                    qa = engine.build_queries(TT, comb, s, p, o if comb == "sp_po" else None, flags=fl)
                    # scoring launch alone (queries prepared once, no next batch)
                    res[f"{tag}prepared_{comb}_us"] = ev(lambda: engine.score_queries(TT, qa, out=out), a.steps)
                    # the pipeline: every launch also builds the next batch's queries
                    pipe = engine.ScorePipeline(TT, comb, n, flags=fl)
                    pipe.start(*batches[0])
                    k = [0]

                    def step():
                        k[0] += 1
                        pipe.step(next_batch=batches[k[0] & 1], out=out)
                    res[f"{tag}pipeline_{comb}_us"] = ev(step, a.steps)
                    res[f"{tag}build_{comb}_us"] = ev(lambda: engine.build_queries(TT, comb, s, p, o if comb == "sp_po" else None, out=qa), a.steps)
                    ab = alg_bytes(n, E, D, sides)
                    res[f"{tag}pipeline_{comb}_frac"] = ab / (res[f"{tag}pipeline_{comb}_us"] * 1e-6) / 8e12
                res[f"split_onecall_{comb}_us"] = ev((lambda: engine.score_sp(TS, s, p)) if comb == "sp_" else
                                                     (lambda: engine.score_sp_po(TS, s, p, o)), a.steps)
            res["coop_one_sided_frac"] = alg_bytes(n, E, D, 1) / (res["coop_one_sided_us"] * 1e-6) / 8e12
            res["coop_two_sided_frac"] = alg_bytes(n, E, D, 2) / (res["coop_two_sided_us"] * 1e-6) / 8e12
            print(json.dumps(res), flush=True)
            del out1, out2
    os.environ.pop("KGE_V4_STORE_SC1", None)
    stamps(512)


def stamps(n):
    """Phase stamps of the prepared one-sided launch (kge_debug_score_sp_bf16_v2, mode 100 / 101 = split)."""
    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    s = torch.randint(E, (n,), generator=g).to(dev)
    p = torch.randint(R, (n,), generator=g).to(dev)
    L = _lib.lib()
    fn = L.kge_debug_score_sp_bf16_v2
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(_lib.KgeTables), _lib.KgeIndex, _lib.KgeIndex, ctypes.c_int64, ctypes.c_int64,
                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                   ctypes.c_void_p]
    for mode, name in ((100, "prepared queries"), (101, "prepared split queries")):
        T = engine.Tables("complex", ent, rel, flags=engine.FLAG_SPLIT_QUERY if mode == 101 else 0)
        tc = T.c()
        keep = []
        si, pi = engine._index(s, dev, keep), engine._index(p, dev, keep)
        out = torch.empty(n, E, device=dev)
        st = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
        wsb = max(int(L.kge_score_workspace_bytes(ctypes.byref(tc), n)), 1 << 20)
        ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
        for _ in range(5):
            st.zero_()
            rc = fn(ctypes.byref(tc), si, pi, n, E, out.data_ptr(), E, st.data_ptr(), mode, ws.data_ptr(), wsb,
                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            assert rc == 0, rc
        v = st.view(4096, 64).cpu()
        v = v[v[:, 0] != 0]
        ld = (v[:, 32:40] - v[:, :1]).double().median(dim=0).values
        print(f"  tile 0: wave 4 issued {float(ld[0]):.0f} landed {float(ld[5]):.0f}; wave 6 issued {float(ld[6]):.0f} landed {float(ld[7]):.0f}")
        print(f"  loader side: first tiles issued {float(ld[0]):.0f}, first store issued {float(ld[1]):.0f} (unit kernel only), store waves' last store issued {float(ld[2]):.0f}, "
              f"acknowledged {float(ld[3]):.0f}, DMA waves' last-tile stores issued {float(ld[4]):.0f}")
        if bool((v[:, 41] != 0).any()):
            arr = (v[:, 40:56] - v[:, :1]).double().median(dim=0).values
            print("  arrival at R(k), DMA wave 4:  " + " ".join(f"{float(x):.0f}" for x in arr[:8]))
            print("  arrival at R(k), store wave 6: " + " ".join(f"{float(x):.0f}" for x in arr[8:]))
        v[:, 32:] = 0
        nst = int((v[0] != 0).sum())
        own = (v[:, :nst] - v[:, :1]).double()
        print(f"==== stamps, {name}, n={n}: {v.shape[0]} workgroups; median cycles since the workgroup's start")
        print("  " + " ".join(f"{float(x):.0f}" for x in own.median(dim=0).values))
        span = (v[:, :nst].max() - v[:, 0].min())
        print(f"  first start -> last stamp: {int(span)} cycles")


if __name__ == "__main__":
    main()
