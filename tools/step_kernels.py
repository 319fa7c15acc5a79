#!/usr/bin/env python3
"""Kernel-level breakdown of the fully fused 1vsAll training step (loss_sp_po + backward + one-pass
Adagrad with bf16 copies) at the C2 shape: run under `rocprofv3 --kernel-trace --stats`."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import model as km, optim as ko

dev = torch.device("cuda", 0)
E, R, d, n = 14541, 237, 512, int(os.environ.get("N", "512"))
g = torch.Generator().manual_seed(0)
s = torch.randint(E, (n,), generator=g).to(dev); p = torch.randint(R, (n,), generator=g).to(dev)
o = torch.randint(E, (n,), generator=g).to(dev)
torch.manual_seed(0)
m = km.create(os.environ.get("MODEL", "complex"), E, R, d, device=dev, score_dtype=torch.bfloat16)
opt = ko.Adagrad(m.parameters(), lr=0.1, bf16_copies=True)
MODE = os.environ.get("MODE", "1vsAll")  # KvsAll: kl (label smoothing LS) and bce losses on random multi-label rows
if MODE != "1vsAll":
    import numpy as np
    rng = np.random.default_rng(0)
    cnt = rng.integers(1, 8, n)
    col = torch.from_numpy(np.concatenate([np.sort(rng.choice(E, c, replace=False)) for c in cnt]).astype(np.int64)).to(dev)
    rowptr = torch.from_numpy(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)).to(dev)
    LS = float(os.environ.get("LS", "0"))
for it in range(int(os.environ.get("STEPS", "30"))):
    opt.zero_grad(set_to_none=True)
    if MODE == "1vsAll":
        m.loss_sp_po(s, p, o).sum().backward()
    elif MODE == "KvsAll":
        m.kl_loss_sp(s, p, rowptr, col, LS).sum().backward()
        m.kl_loss_po(p, o, rowptr, col, LS).sum().backward()
    else:
        m.bce_loss_sp(s, p, rowptr, col, 0.0, LS).sum().backward()
        m.bce_loss_po(p, o, rowptr, col, 0.0, LS).sum().backward()
    opt.step()
torch.cuda.synchronize()
print("done")
