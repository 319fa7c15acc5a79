"""bench.py's roofline_rank legs alone (the counting kernel in its three forms: split, single-pass, band-and-rescore)
   python tools/rank_band_bench.py [steps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kge_amd import engine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
out = bench.rank_legs(engine, dev, bench.BATCH, steps)
for tag in ("fb15k-237", "wikidata5m_shard"):
    leg = out[tag]
    print(tag, json.dumps({k: leg[k] for k in ("planted_true_scores",)}, indent=1))
    print(tag, "random-table legs: parity fused_us", leg["parity"]["fused_us"], "single-pass fused_us",
          leg["training_tolerance"]["fused_us"])

# ---- what the band form costs WITHOUT any rescoring (true scores far above every score: nothing inside a band) and the
# share of its time that is the rare path, at the Wikidata5M-shard shape
import numpy as np
E, R, d, n = (bench.E_WD + 7) // 8, bench.R_WD, bench.DIM_WD, bench.BATCH
g = torch.Generator(device=dev).manual_seed(7)
ent = (torch.randn(E, d, generator=g, device=dev) * 0.3).bfloat16()
rel = (torch.randn(R, d, generator=g, device=dev) * 0.3).bfloat16()
rng = np.random.default_rng(0)
s, p, o = (torch.from_numpy(rng.integers(0, hi, n)).to(dev) for hi in (E, R, E))
far = torch.full((n,), 1.0e4, device=dev)
cnt = torch.zeros(2, 2, 1, n, dtype=torch.int64, device=dev)
Tsp = engine.Tables("complex", ent, rel, flags=engine.FLAG_SPLIT_QUERY)
T1 = engine.Tables("complex", ent, rel, flags=0)
band = engine.RankBand(Tsp, n)
for tag, T_, b in (("split", Tsp, None), ("single_pass", T1, None), ("band_nothing_to_rescore", Tsp, band)):
    fn = lambda: engine.score_rank_sp_po(T_, s, p, o, far, far, [], [], 1e-5, 1e-4, cnt[0, 0], cnt[0, 1], cnt[1, 0],
                                         cnt[1, 1], band=b)
    for _ in range(3):
        fn()
    print("far true scores, no filters:", tag, "%.1f us" % (bench.event_avg_ms(fn, steps) * 1e3))
print("pairs listed, dropped:", band.status())

# the same at the FB15k-237 shape (d = 512): where the band form's fixed costs show
E, R, d = bench.E_FB, bench.R_FB, bench.DIM
g = torch.Generator(device=dev).manual_seed(7)
ent = (torch.randn(E, d, generator=g, device=dev) * 0.3).bfloat16()
rel = (torch.randn(R, d, generator=g, device=dev) * 0.3).bfloat16()
s, p, o = (torch.from_numpy(rng.integers(0, hi, n)).to(dev) for hi in (E, R, E))
Tsp = engine.Tables("complex", ent, rel, flags=engine.FLAG_SPLIT_QUERY)
T1 = engine.Tables("complex", ent, rel, flags=0)
band = engine.RankBand(Tsp, n)
for tag, T_, b in (("split", Tsp, None), ("single_pass", T1, None), ("band_nothing_to_rescore", Tsp, band)):
    fn = lambda: engine.score_rank_sp_po(T_, s, p, o, far, far, [], [], 1e-5, 1e-4, cnt[0, 0], cnt[0, 1], cnt[1, 0],
                                         cnt[1, 1], band=b)
    for _ in range(3):
        fn()
    print("FB15k-237 shape, far true scores, no filters:", tag, "%.1f us" % (bench.event_avg_ms(fn, steps) * 1e3))
