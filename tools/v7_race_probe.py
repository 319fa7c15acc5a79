"""Stress of ScorePipeline(streams=3) on the v7 kernel: where do mismatches against the one-call scores fall?"""
import sys, collections
import numpy as np
import torch
sys.path.insert(0, ".")
from kge_amd import engine as eng
dev = "cuda:0"
E, R, d, n = 14541, 237, 512, 512
g = torch.Generator().manual_seed(6)
ent = (torch.randn(E, d, generator=g) * 0.3).bfloat16().to(dev)
rel = (torch.randn(R, d, generator=g) * 0.3).bfloat16().to(dev)
T = eng.Tables("complex", ent, rel)
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nb = 7
trip = []
for k in range(nb):
    q = torch.Generator().manual_seed(20 + k)
    trip.append(torch.stack([torch.randint(hi, (n,), generator=q) for hi in (E, R, E)], 1).to(dev))
want = [eng.score_sp(T, t[:, 0], t[:, 1]) for t in trip]
torch.cuda.synchronize()
bad_total = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    pipe = eng.ScorePipeline(T, "sp_", n, streams=lanes)
    outs = [torch.full_like(want[0], float("nan")) for _ in range(lanes)]
    pipe.start(trip[:lanes])
    got = []
    for k in range(nb):
        pipe.step(next_batch=trip[k + lanes] if k + lanes < nb else None, out=outs[k % lanes])
        if k % lanes == lanes - 1 or k == nb - 1:
            pipe.join()
            for j in range(k - k % lanes, k + 1):
                got.append(outs[j % lanes].clone())
                outs[j % lanes].fill_(float("nan"))
            pipe.fork()
    torch.cuda.synchronize()
    for k in range(nb):
        a, b = got[k].cpu().numpy(), want[k].cpu().numpy()
        bad = a != b
        if bad.any():
            rc = np.argwhere(bad)
            rows = collections.Counter(rc[:, 0].tolist())
            cols = rc[:, 1]
            nan = int(np.isnan(a[bad]).sum())
            bad_total += len(rc)
            print(f"iter {it} batch {k}: {len(rc)} bad ({nan} still NaN = never written); rows {dict(rows)}; "
                  f"cols {cols.min()}..{cols.max()}; col%32 {sorted(set((cols % 32).tolist()))[:40]}; "
                  f"units {sorted(set((cols // 32).tolist()))[:20]}", flush=True)
print("total bad", bad_total)
