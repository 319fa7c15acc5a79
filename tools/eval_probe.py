#!/usr/bin/env python3
"""EntityRankingJob-equivalent evaluation (kge_amd.eval.EntityRankingEvaluator) at the C4 shape:
E=14,541, R=237, d=512 DistMult bf16, 272,115 synthetic train / 17,535 valid / 20,466 test triples
(Zipf entity popularity), batch 512: wall time, and host vs device split."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine, eval as kev
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _eval_env  # KGE_EVAL_* variables -> EntityRankingEvaluator.OPTIONS
_eval_env.apply()

dev = torch.device("cuda", 0)
E, R, d = 14541, 237, 512
rng = np.random.default_rng(0)


def zipf_triples(k):
    w = 1.0 / np.arange(1, E + 1); w /= w.sum()
    return np.stack([rng.choice(E, k, p=w), rng.integers(0, R, k), rng.choice(E, k, p=w)], 1).astype(np.int64)


splits = {"train": zipf_triples(272115), "valid": zipf_triples(17535), "test": zipf_triples(20466)}
g = torch.Generator().manual_seed(0)
ent = torch.empty(E, d).normal_(0, 0.1, generator=g).bfloat16().to(dev)
rel = torch.empty(R, d).normal_(0, 0.1, generator=g).bfloat16().to(dev)
T = engine.Tables("distmult", ent, rel)
bs = int(os.environ.get("BS", "512"))
t0 = time.perf_counter()
ev = kev.EntityRankingEvaluator(T, splits, E, R, batch_size=bs)
t1 = time.perf_counter()
m = ev.run(); torch.cuda.synchronize()
t2 = time.perf_counter()
m = ev.run(); torch.cuda.synchronize()
t3 = time.perf_counter()
nb = (len(splits["valid"]) + bs - 1) // bs
print(f"index build {t1-t0:.2f} s; eval of {len(splits['valid'])} triples: first {t2-t1:.3f} s, second {t3-t2:.3f} s "
      f"= {1e3*(t3-t2)/nb:.2f} ms per batch of {bs}; MRR filt {m['mean_reciprocal_rank_filtered']:.5f}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); ev.run(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
