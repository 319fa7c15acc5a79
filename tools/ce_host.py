#!/usr/bin/env python3
"""Host issue time per engine call (no device waits inside the loops; queue drained between)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
E, R, d, n = 14541, 237, 512, 512
g = torch.Generator().manual_seed(0)
T = engine.Tables("complex", torch.empty(E, d).normal_(0, 0.1, generator=g).bfloat16().to(dev),
                  torch.empty(R, d).normal_(0, 0.1, generator=g).bfloat16().to(dev))
s, p, o = (torch.randint(k, (n,), generator=g).to(dev) for k in (E, R, E))
loss, lse = engine.ce_fwd(T, "sp", s, p, o)
sc = engine.score_sp(T, s, p)
ds = torch.softmax(sc, 1)
gr = torch.full((n,), 1.0 / n, device=dev)


def host(fn, k=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / k, 1e6 * (t2 - t0) / k


for nm, fn in (("engine.score_sp", lambda: engine.score_sp(T, s, p)),
               ("engine.ce_fwd", lambda: engine.ce_fwd(T, "sp", s, p, o)),
               ("engine.ce_bwd", lambda: engine.ce_bwd(T, "sp", s, p, o, lse, g_rows=gr)),
               ("engine.score_pairs_bwd", lambda: engine.score_pairs_bwd(T, "sp", s, p, None, ds)),
               ("torch.softmax", lambda: torch.softmax(sc, 1)),
               ("torch.empty+fill", lambda: torch.zeros(E, d, device=dev))):
    h, t = host(fn)
    print(f"{nm:26s} host issue {h:7.1f} us/call   wall {t:7.1f} us/call")
