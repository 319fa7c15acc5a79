#!/usr/bin/env python3
"""EntityRankingEvaluator at the C4 shape on float32 tables (the default precision of a LibKGE model): the counting
epilogue of the exact kernels (kge_eval_batch) against the two-step loop (score matrix + scans), per scorer."""
import os, sys, time
os.environ["KGE_EVAL_FUSED_EXACT"] = "1"
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine, eval as kev
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _eval_env  # KGE_EVAL_* variables -> EntityRankingEvaluator.OPTIONS
_eval_env.apply()
from kge_amd.synthetic import make_splits
dev = torch.device("cuda", 0)
E, R, d, bs = 14541, 237, 512, 512
splits = {k: v.astype(np.int64) for k, v in make_splits(E, R, 272115, 17535, 20466, seed=0).items()}
g = torch.Generator().manual_seed(0)
for model, dd in (("distmult", 512), ("complex", 512), ("transe", 128), ("rotate", 128)):
    ent = torch.empty(E, dd).normal_(0, 0.1, generator=g).to(dev)
    rel = torch.empty(R, dd // 2 if model == "rotate" else dd).normal_(0, 0.1, generator=g).to(dev)
    T = engine.Tables(model, ent, rel)
    res = {}
    for tag, fused in (("counting", True), ("two_step", False)):
        ev = kev.EntityRankingEvaluator(T, splits, E, R, batch_size=bs)
        ev._fused = fused
        ev.run(); torch.cuda.synchronize()
        t0 = time.perf_counter(); m = ev.run(); torch.cuda.synchronize()
        res[tag] = (time.perf_counter() - t0) * 1e3 / ((17535 + bs - 1) // bs)
        res[tag + "_mrr"] = m["mean_reciprocal_rank_filtered"]
    print(f"{model:9s} d={dd}: counting kernels {res['counting']:.3f} ms/batch, two-step {res['two_step']:.3f} ms/batch; "
          f"MRR {res['counting_mrr']:.6f} / {res['two_step_mrr']:.6f}", flush=True)
