"""What the filter lists cost a counting call at the Wikidata5M-shard shape: K = 0 / 1 / 2 filter sets, split and single-pass
   python tools/bits_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kge_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
E, R, d, n = (bench.E_WD + 7) // 8, bench.R_WD, bench.DIM_WD, bench.BATCH
g = torch.Generator(device=dev).manual_seed(7)
ent = (torch.randn(E, d, generator=g, device=dev) * 0.3).bfloat16()
rel = (torch.randn(R, d, generator=g, device=dev) * 0.3).bfloat16()
rng = np.random.default_rng(0)
s, p, o = (torch.from_numpy(rng.integers(0, hi, n)).to(dev) for hi in (E, R, E))
far = torch.full((n,), 1.0e4, device=dev)
lists = []
for tc in (o.cpu().numpy(), s.cpu().numpy()):
    per = [np.unique(np.append(rng.integers(0, E, 4), c)) for c in tc]
    end = np.cumsum([len(x) for x in per])
    beg = end - np.array([len(x) for x in per])
    one = tuple(torch.from_numpy(np.asarray(x, np.int64)).to(dev) for x in (beg, end, np.concatenate(per)))
    lists.append([one, one])
for flags, tag in ((engine.FLAG_SPLIT_QUERY, "split"), (0, "single-pass")):
    T = engine.Tables("complex", ent, rel, flags=flags)
    for K in (0, 1, 2):
        cnt = torch.zeros(2, 2, K + 1, n, dtype=torch.int64, device=dev)
        fn = lambda: engine.score_rank_sp_po(T, s, p, o, far, far, lists[0][:K], lists[1][:K], 1e-5, 1e-4, cnt[0, 0], cnt[0, 1],
                                             cnt[1, 0], cnt[1, 1])
        for _ in range(3):
            fn()
        print(f"{tag}, {K} filter sets: {bench.event_avg_ms(fn, 20) * 1e3:.1f} us per call")
