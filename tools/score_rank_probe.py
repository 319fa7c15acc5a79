"""Time the fused score + rank call against the two-step path at the C4 evaluation shape (run under
rocprofv3 --kernel-trace --stats for the per-kernel split):  python tools/score_rank_probe.py [n] [E] [d]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kge_amd import engine as eng  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
E = int(sys.argv[2]) if len(sys.argv) > 2 else 14541
d = int(sys.argv[3]) if len(sys.argv) > 3 else 512
R, dev = 237, torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
T = eng.Tables("complex", (torch.randn(E, d, generator=g) * 0.3).bfloat16().to(dev),
               (torch.randn(R, d, generator=g) * 0.3).bfloat16().to(dev), 1.0)
rng = np.random.default_rng(0)
s, p, o = (torch.from_numpy(rng.integers(0, hi, n)).to(dev) for hi in (E, R, E))
t_sp = eng.score_sp(T, s, p, o).diagonal().contiguous()
t_po = eng.score_po(T, p, o, s).diagonal().contiguous()


def filt(tc):
    b, e, v = np.zeros(n, np.int64), np.zeros(n, np.int64), []
    for i in range(n):
        b[i] = len(v)
        v.extend(np.unique(np.append(rng.integers(0, E, 4), tc[i])).tolist())
        e[i] = len(v)
    return tuple(torch.from_numpy(np.asarray(x, np.int64)).to(dev) for x in (b, e, v))


f_sp = [filt(o.cpu().numpy()), filt(o.cpu().numpy())]
f_po = [filt(s.cpu().numpy()), filt(s.cpu().numpy())]
cnt = torch.zeros(2, 2, 3, n, dtype=torch.int64, device=dev)
oc, sc_ = o.contiguous(), s.contiguous()


def fused():
    eng.score_rank_sp_po(T, s, p, o, t_sp, t_po, f_sp, f_po, 1e-5, 1e-4, cnt[0, 0], cnt[0, 1], cnt[1, 0], cnt[1, 1])


def two_step():
    sc = eng.score_sp_po(T, s, p, o)
    eng.rank_counts_multi(sc[:, :E], t_sp, f_sp, 0, oc, 1e-5, 1e-4, cnt[0, 0], cnt[0, 1])
    eng.rank_counts_multi(sc[:, E:], t_po, f_po, 0, sc_, 1e-5, 1e-4, cnt[1, 0], cnt[1, 1])


for name, fn in (("two-step", two_step), ("fused", fused)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100)
    print(f"{name}: {best * 1e6:.1f} us per batch (n={n}, E={E}, d={d}, wall clock incl. host issue)")
