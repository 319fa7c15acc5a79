"""bench.py's eval_leg with 1 and 2 captured lanes (KGE_EVAL_LANES): ms per batch of a whole evaluation pass."""
import os, sys
import torch
sys.path.insert(0, ".")
import bench
from kge_amd import engine
for L in ("1", "2", "3"):
    from kge_amd.eval import EntityRankingEvaluator
    EntityRankingEvaluator.OPTIONS["lanes"] = int(L)
    r = bench.eval_leg(engine, torch.device("cuda:0"))
    print(L, {k: (round(v["ms_per_batch"], 4), v["graph_batches"], round(v["mrr_filtered"], 8)) for k, v in r.items() if isinstance(v, dict)}, flush=True)
