#!/usr/bin/env python3
"""pairs_bf16_v7_kernel (direct stores) against pairs_bf16_v6_kernel (KGE_V7=0; earlier: v6 against v4, KGE_V6=0) on prepared queries at the FB15k-237 shape: the
pipelined step (every launch also builds the next batch's queries), one- and two-sided, contiguous and padded
pitch, plain and split queries; variants timed in turn (ABAB...), median of R rounds of S back-to-back steps.
Then the phase stamps of the one-sided launch.      python tools/v6_probe.py [--steps 300] [--rounds 5]"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kge_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
E, R, D = 14541, 237, 512
VARIANTS = {"v7": {"KGE_V6": "1", "KGE_V7": "1"}, "v6": {"KGE_V6": "1", "KGE_V7": "0"}}

def alg_bytes(n, m, d, sides):
    return m * d * 2 + sides * (n * 2 * d * 2 + n * m * 4 + 2 * n * 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--stamps", type=int, default=1)
    ap.add_argument("--two-sided-stamps", action="store_true")
    a = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    cases = []
    for n in (512, 1024):
        batches = [tuple(torch.randint(hi, (n,), generator=g).to(dev) for hi in (E, R, E)) for _ in range(2)]
        for comb, sides in (("sp_", 1), ("sp_po", 2)):
            for split in (0,):
                fl = engine.FLAG_SPLIT_QUERY if split else None
                T = engine.Tables("complex", ent, rel, flags=fl or 0)
                pipe = engine.ScorePipeline(T, comb, n, flags=fl)
                pipe.start(*batches[0])
                for pad in (0, 1):
                    P = engine.score_pitch(E) if pad else E
                    buf = torch.empty(n, sides * P, device=dev)
                    out = buf.view(n, sides, P)[:, :, :E] if sides == 2 else buf[:, :E]
                    if sides == 2 and not pad:
                        out = buf
                    cases.append(dict(n=n, combine=comb, split=split, pad=pad, pipe=pipe, out=out, batches=batches,
                                      bytes=alg_bytes(n, E, D, sides), t={v: [] for v in VARIANTS}))
    k = [0]
    for r in range(a.rounds):
        for c in cases:
            for v6 in VARIANTS:
                os.environ.update(VARIANTS[v6])

                def step():
                    k[0] += 1
                    c["pipe"].step(next_batch=c["batches"][k[0] & 1], out=c["out"])
                for _ in range(20):
                    step()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(a.steps):
                    step()
                e1.record()
                torch.cuda.synchronize()
                c["t"][v6].append(e0.elapsed_time(e1) / a.steps * 1e3)
    for c in cases:
        row = {"n": c["n"], "combine": c["combine"], "split": c["split"], "padded_pitch": c["pad"]}
        for v in VARIANTS:
            md = statistics.median(c["t"][v])
            row[v + "_us"] = round(md, 2)
            row[v + "_frac"] = round(c["bytes"] / (md * 1e-6) / 8e12, 3)
        print(json.dumps(row), flush=True)
    if a.stamps:
        import prep_probe
        for v6 in VARIANTS:
            os.environ.update(VARIANTS[v6])
            print(f"######## {v6}")
            prep_probe.stamps(512)


if __name__ == "__main__" and "--two-sided-stamps" not in sys.argv:
    main()


def stamps_two_sided(n=512, pitch=14656):
    """Phase stamps of the pipelined two-sided launch (kge_debug_v6_stamps)."""
    import ctypes
    from kge_amd import _lib
    L = _lib.lib()
    L.kge_debug_v6_stamps.restype = None
    L.kge_debug_v6_stamps.argtypes = [ctypes.c_void_p]
    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    T = engine.Tables("complex", ent, rel)
    batches = [tuple(torch.randint(hi, (n,), generator=g).to(dev) for hi in (E, R, E)) for _ in range(2)]
    pipe = engine.ScorePipeline(T, "sp_po", n)
    pipe.start(*batches[0])
    buf = torch.empty(n, 2 * pitch, device=dev)
    out = buf.view(n, 2, pitch)[:, :, :E]
    st = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
    for k in range(6):
        pipe.step(next_batch=batches[k & 1], out=out)
    torch.cuda.synchronize()
    st.zero_()
    L.kge_debug_v6_stamps(ctypes.c_void_p(st.data_ptr()))
    pipe.step(next_batch=batches[0], out=out)
    torch.cuda.synchronize()
    L.kge_debug_v6_stamps(None)
    v = st.view(4096, 64).cpu()
    v = v[(v[:, 0] != 0) & (v[:, 3] != 0)]
    ld = (v[:, 32:56] - v[:, :1]).double().median(dim=0).values
    print(f"==== two-sided pipelined launch, n={n}, pitch {pitch}: {v.shape[0]} scoring workgroups")
    print(f"  first units issued {float(ld[0]):.0f}, unit 0 landed {float(ld[5]):.0f}, first stores issued {float(ld[1]):.0f}, "
          f"last store issued {float(ld[2]):.0f}, acknowledged {float(ld[3]):.0f}, DMA waves' last-unit stores {float(ld[4]):.0f}")
    print("  arrival at P(u), DMA wave 4:   " + " ".join(f"{float(x):.0f}" for x in ld[8:16]))
    print("  arrival at P(u), store wave 6: " + " ".join(f"{float(x):.0f}" for x in ld[16:24]))
    v[:, 32:] = 0
    nst = int((v[0] != 0).sum())
    own = (v[:, :nst] - v[:, :1]).double()
    print("  consumer wave 0: " + " ".join(f"{float(x):.0f}" for x in own.median(dim=0).values))
    span = (v[:, :nst].max() - v[:, 0].min())
    print(f"  first start -> last consumer stamp: {int(span)} cycles; starts spread over {int(v[:, 0].max() - v[:, 0].min())} cycles")


if __name__ == "__main__" and "--two-sided-stamps" in sys.argv:
    stamps_two_sided()
