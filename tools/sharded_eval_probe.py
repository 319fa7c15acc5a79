#!/usr/bin/env python3
"""One evaluation batch on ONE rank's shard of the Wikidata5M shape (574,311 entity rows = E / 8, R = 822, d = 256,
batch 512; raw + two filtered rankings, both directions) through ShardedEntityTable.rank_batch_multi -- what each of 8
ranks does per batch -- with a one-rank RCCL group running every collective (row all-gather, counter all-reduce):
counts inside the scoring kernel vs score slabs + rank_counts_multi.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 tools/sharded_eval_probe.py"""
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd.sharded import ShardedEntityTable

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if "RANK" in os.environ:
    dist.init_process_group("nccl", device_id=dev)
Eg, R, d, n = int(os.environ.get("ROWS", "574311")), 822, 256, int(os.environ.get("BS", "512"))
g = torch.Generator(device=dev).manual_seed(0)
ent = torch.empty(Eg, d, device=dev).normal_(0, 0.1, generator=g).bfloat16()
rel = torch.empty(R, d, device=dev).normal_(0, 0.1, generator=g).bfloat16()
sh = ShardedEntityTable("complex", ent, rel, Eg, force_collectives=dist.is_initialized())
rng = np.random.default_rng(0)
tri = torch.from_numpy(np.stack([rng.integers(0, Eg, n), rng.integers(0, R, n), rng.integers(0, Eg, n)], 1)).to(dev)


def filt(tc):
    per = [np.unique(np.append(rng.integers(0, Eg, 4), c)) for c in tc]
    end = np.cumsum([len(x) for x in per])
    beg = end - np.array([len(x) for x in per])
    return tuple(torch.from_numpy(np.asarray(x, np.int64)).to(dev) for x in (beg, end, np.concatenate(per)))


fo = [filt(tri[:, 2].cpu().numpy())] * 2
fs = [filt(tri[:, 0].cpu().numpy())] * 2
res = {}
for name, fused in (("counts inside the scoring kernel", True), ("score slabs + rank_counts_multi", False)):
    sh.fused_rank = fused
    for _ in range(3):
        c = sh.rank_batch_multi(tri, fo, fs)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            c = sh.rank_batch_multi(tri, fo, fs)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    res[name] = c.clone()
    print(f"{name}: {best * 1e3:.3f} ms per batch of {n} on a {Eg}-row shard (collectives: {sh.collectives})", flush=True)
a, b = res.values()
print("counts identical:", bool(torch.equal(a, b)))
if dist.is_initialized():
    dist.destroy_process_group()
