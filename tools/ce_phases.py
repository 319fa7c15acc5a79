#!/usr/bin/env python3
"""s_memtime stamps inside the fused-loss scoring kernels (pairs_bf16_v3_kernel<.., V3_LSE / V3_DS>)
at the BASELINE configs[1] shape (or E= D= N= from the environment, e.g. E=574311 D=256: a Wikidata5M shard): median over workgroups of each stamp relative to the workgroup's
own start (kge_debug_ce_stamps, not part of the ABI)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import _lib, engine  # noqa: E402

dev = torch.device("cuda", 0)
E, R, d, n = int(os.environ.get("E", "14541")), 237, int(os.environ.get("D", "512")), int(os.environ.get("N", "512"))
g = torch.Generator().manual_seed(0)
T = engine.Tables("complex", torch.empty(E, d).normal_(0, 0.1, generator=g).bfloat16().to(dev),
                  torch.empty(R, d).normal_(0, 0.1, generator=g).bfloat16().to(dev))
s, p, o = (torch.randint(k, (n,), generator=g).to(dev) for k in (E, R, E))
L = _lib.lib()
L.kge_debug_ce_stamps.restype = None
L.kge_debug_ce_stamps.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
loss, lse = engine.ce_fwd(T, "sp", s, p, o)
names = ["start", "share built+published", "tiles 0,1 issued", "flags seen", "fragments loaded"]
for what in ("fwd (V3_LSE)", "bwd (V3_DS)"):
    for _ in range(3):
        stamps.zero_()
        L.kge_debug_ce_stamps(stamps.data_ptr())
        if what.startswith("fwd"):
            engine.ce_fwd(T, "sp", s, p, o)
        else:
            engine.ce_bwd(T, "sp", s, p, o, lse, g_scalar=1.0 / n)
        torch.cuda.synchronize()
        L.kge_debug_ce_stamps(None)
    st = stamps.view(4096, 64).cpu()
    st = st[st[:, 0] != 0]
    nst = int((st[0] != 0).sum())
    own = (st[:, :nst] - st[:, :1]).double()
    med = own.median(dim=0).values
    print(f"==== {what}: {st.shape[0]} workgroups, {nst} stamps")
    nm = list(names)
    tt = 0
    while len(nm) < nst:
        nm += [f"tile{tt} released", f"tile{tt} mfma issued"]
        tt += 1
    prev = 0.0
    for i in range(nst):
        print(f"  {i:2d} {nm[i]:<24s} median {float(med[i]):8.0f}  (+{float(med[i]) - prev:6.0f})")
        prev = float(med[i])
