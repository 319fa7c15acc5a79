#!/usr/bin/env python3
"""30 calls of kge_score_rank_sp_po on one Wikidata5M shard (n = 512, E = 574,311, d = 256, ComplEx, no filters):
the workload of the counter passes over the counting kernel (tools/rank_pmc.sh)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kge_amd import engine as eng
n, E, R, d, dev = 512, 574311, 822, 256, torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
T = eng.Tables("complex", (torch.randn(E, d, generator=g, device=dev) * 0.3).bfloat16(),
               (torch.randn(R, d, generator=g, device=dev) * 0.3).bfloat16())
rng = np.random.default_rng(0)
s, p, o = (torch.from_numpy(rng.integers(0, hi, n)).to(dev) for hi in (E, R, E))
t_sp = eng.score_sp(T, s, p, o).diagonal().contiguous()
t_po = eng.score_po(T, p, o, s).diagonal().contiguous()
cnt = torch.zeros(2, 2, 1, n, dtype=torch.int64, device=dev)
for _ in range(30):
    assert eng.score_rank_sp_po(T, s, p, o, t_sp, t_po, [], [], 1e-5, 1e-4, cnt[0, 0], cnt[0, 1], cnt[1, 0], cnt[1, 1])
torch.cuda.synchronize()
