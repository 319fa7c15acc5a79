#!/usr/bin/env python3
"""KGE_ROCTX=1: a few scoring calls; run under `rocprofv3 --marker-trace --kernel-trace` the trace must carry one roctx
range per C entry point (include/kge_amd.h), named after it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kge_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
ent = torch.randn(3000, 512, generator=g).bfloat16().to(dev)
rel = torch.randn(7, 512, generator=g).bfloat16().to(dev)
T = engine.Tables("complex", ent, rel)
s, p, o = (torch.randint(hi, (256,), generator=g).to(dev) for hi in (3000, 7, 3000))
for _ in range(3):
    engine.score_sp(T, s, p)
    engine.score_sp_po(T, s, p, o)
    engine.score_spo(engine.Tables("complex", ent.float(), rel.float()), s, p, o)
torch.cuda.synchronize()
print("done")
