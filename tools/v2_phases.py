#!/usr/bin/env python3
"""Per-phase timestamps of the row-persistent bf16 kernel (kge_debug_score_sp_bf16_v2):
every workgroup's wave 0 records s_memtime at fixed points; this prints, relative to the
earliest kernel-start stamp, the median / max over workgroups of each stamp."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import _lib, engine  # noqa: E402

dev = torch.device("cuda", 0)


def run(n, E=14541, R=237, d=512, reps=5, mode=0, use_ws=False, flags=0):
    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E, d).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    rel = torch.empty(R, d).normal_(0, 0.1, generator=g).bfloat16().to(dev)
    s = torch.randint(E, (n,), generator=g).to(dev)
    p = torch.randint(R, (n,), generator=g).to(dev)
    T = engine.Tables("complex", ent, rel, flags=flags)
    out = torch.empty(n, E, device=dev)
    L = _lib.lib()
    fn = L.kge_debug_score_sp_bf16_v2
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(_lib.KgeTables), _lib.KgeIndex, _lib.KgeIndex, ctypes.c_int64,
                   ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    keep = []
    si, pi = engine._index(s, dev, keep), engine._index(p, dev, keep)
    nwg = 4096
    stamps = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)
    tc = T.c()
    wsb = max(int(L.kge_score_workspace_bytes(ctypes.byref(tc), n)), 1 << 20)
    ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
    for _ in range(reps):
        stamps.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(ctypes.byref(tc), si, pi, n, E, out.data_ptr(), E, stamps.data_ptr(), mode, ws.data_ptr() if use_ws else None, wsb if use_ws else 0,
                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        b.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
    st = stamps.view(nwg, 64).cpu()
    ld = st[:, 32] != 0
    if bool(ld.any()):
        bb = st[ld]
        nm4 = ["landed", "B1 passed", "dma issued", "stores issued"]
        for c in range(32, 64):
            if not bool((bb[:, c] != 0).all()):
                break
            d = (bb[:, c] - bb[:, 0]).double()
            print(f"  loader step {(c - 32) // 4} {nm4[(c - 32) % 4]:<14s} median {float(d.median()):8.0f}")
    st[:, 32:60] = 0
    bld = st[:, 60] != 0
    if bool(bld.any()):
        bb = st[bld]
        for nm, col in (("idx loaded", 60), ("rows gathered", 61), ("stores acked", 62)):
            d = (bb[:, col] - bb[:, 0]).double()
            print(f"  builder ({int(bld.sum())} wgs) {nm:<14s} median {float(d.median()):8.0f}  min {float(d.min()):8.0f}  max {float(d.max()):8.0f}")
    st[:, 60:] = 0
    used = st[:, 0] != 0
    st = st[used]
    t0 = st[:, 0].min()
    nst = int((st[0] != 0).sum())
    rel_t = (st[:, :nst] - t0).double()
    total = float(rel_t.max())
    us = a.elapsed_time(b) * 1e3
    # per-workgroup relative times (counters are per XCD): median over workgroups of (stamp_i - stamp_0)
    own = (st[:, :nst] - st[:, :1]).double()
    med = own.median(dim=0).values
    print(json.dumps({"mode": mode, "ws": use_ws, "n": n, "own_median": [float(x) for x in med]}))
    print(json.dumps({"n": n, "workgroups": int(used.sum()), "stamps": nst, "event_us": us,
                      "span_ticks": total, "ticks_per_us_if_span_eq_event": total / us}))
    if use_ws and not (flags & 24):
        names = ["start", "share built+published", "flags seen", "fragment loads issued"]
    elif use_ws and not (flags & 8):
        names = ["start", "share built+published", "tiles 0,1 issued", "flags seen", "fragments loaded"]
    else:
        names = ["start", "T0+idx+ptrs", "gathers issued"]
        passes = 4
        for p_ in range(passes):
            names += [f"pass{p_} landed", f"pass{p_} built"]
        names += ["prologue done"]
    k = len(names)
    tt = 0
    while len(names) < nst:
        names += [f"tile{tt} released", f"tile{tt} mfma issued"]
        tt += 1
    for i in range(nst):
        col = rel_t[:, i]
        print(f"  {i:2d} {names[i]:<22s} median {float(col.median()):10.0f}  min {float(col.min()):10.0f}  max {float(col.max()):10.0f}")


if __name__ == "__main__":
    for n, ws in ((128, False), (512, False), (1024, False), (128, True), (512, True), (1024, True)):
        print(f"==== {'v4 (loader/consumer waves)' if ws else 'v3 (64-target tiles)'} n={n} workspace={ws}")
        run(n, use_ws=ws)
    print("==== v3 (single role, cooperative build) n=512 workspace=True")
    run(512, use_ws=True, flags=16)

