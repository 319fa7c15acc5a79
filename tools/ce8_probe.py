#!/usr/bin/env python3
"""Timing of the two fused-loss passes at the FB15k-237 shape (two-sided, n = 512): persistent kernel (CE_V8 = 1)
against the loader/consumer kernel (CE_V8 = 0), HIP events over REPS calls each.  tools/gpu_ce8probe.sh builds the
probe variants (make CXXEXTRA=-DKGE_V8C_PROBE=1: the chains alone)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine, _lib
dev = torch.device("cuda", 0)
E, R, D = int(os.environ.get("E", 14541)), 237, int(os.environ.get("D", 512))
N = int(os.environ.get("N", 512))
REPS = int(os.environ.get("REPS", 200))
g = torch.Generator().manual_seed(0)
ent = (torch.randn(E, D, generator=g) * 0.1).to(dev).bfloat16()
rel = (torch.randn(R, D, generator=g) * 0.1).to(dev).bfloat16()
s, p, o = (torch.randint(h, (N,), generator=g).to(dev) for h in (E, R, E))
T = engine.Tables("complex", ent, rel)
def timeit(fn):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / REPS
for v8 in (1, 0):
    _lib.set_switch("CE_V8", v8)
    loss, lse = engine.ce_sp_po_fwd(T, s, p, o)
    tf = timeit(lambda: engine.ce_sp_po_fwd(T, s, p, o))
    tb = timeit(lambda: engine.ce_sp_po_bwd_accum(T, s, p, o, lse, g_scalar=1.0 / N))
    print(f"CE_V8={v8} N={N} E={E} D={D}: fwd call {tf:.1f} us, bwd call {tb:.1f} us (launch-by-launch, host-issued)", flush=True)
