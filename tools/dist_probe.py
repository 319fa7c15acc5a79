#!/usr/bin/env python3
"""Timing of the pieces of bench.py's sharded step on ONE rank (RCCL initialised): exchange graph,
scoring graph, both on two streams.  Run under torch.distributed.run."""
import os, sys, time
import torch
import torch.distributed as td
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kge_amd import engine

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
td.init_process_group("nccl", device_id=dev)
n, DIM = 512, bench.DIM
ent, rel, s, p, o = bench.make_inputs(0, dev, n)
so_idx = torch.stack([s, o], 1).reshape(-1)
loc = torch.empty(2 * n, DIM, dtype=torch.bfloat16, device=dev)
gath = torch.empty(n, 2 * DIM, dtype=torch.bfloat16, device=dev)
pe = torch.empty(n, DIM, dtype=torch.bfloat16, device=dev)


def exchange():
    torch.index_select(ent, 0, so_idx, out=loc)
    td.all_gather_into_tensor(gath.view(-1), loc.view(-1))
    torch.index_select(rel, 0, p, out=pe)


def score(fl):
    engine.score_emb("complex", gath[:, :DIM], pe, ent, "sp_", flags=fl)
    engine.score_emb("complex", ent, pe, gath[:, DIM:], "_po", flags=fl)


def cap(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        fn()
    g.replay(); torch.cuda.synchronize()
    return g


def timeit(fn, k=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / k, 1e6 * (t2 - t0) / k


gx = cap(exchange)
for R in (0, 8):
    fl = engine.reserve_cus(R)
    gs = cap(lambda: score(fl))
    print(f"reserve {R}: scoring graph      host %.1f us  total %.1f us" % timeit(gs.replay))
print("exchange graph            host %.1f us  total %.1f us" % timeit(gx.replay))
print("exchange eager            host %.1f us  total %.1f us" % timeit(exchange))
g1 = cap(lambda: torch.index_select(ent, 0, so_idx, out=loc))
print("  index_select(ent) graph host %.1f us  total %.1f us" % timeit(g1.replay))
g2 = cap(lambda: td.all_gather_into_tensor(gath.view(-1), loc.view(-1)))
print("  all_gather graph        host %.1f us  total %.1f us" % timeit(g2.replay))
comm = torch.cuda.Stream(dev)
ev1, ev2 = torch.cuda.Event(), torch.cuda.Event()


def both():
    with torch.cuda.stream(comm):
        gx.replay()
    gs.replay()


print("both, two streams, no deps host %.1f us  total %.1f us" % timeit(both))


def host_only(fn, k=2000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return 1e6 * (t1 - t0) / k


cs = torch.cuda.current_stream(dev)
print("host: ev.record(cs)            %.1f us" % host_only(lambda: ev1.record(cs)))
print("host: cs.wait_event            %.1f us" % host_only(lambda: cs.wait_event(ev1)))


def ctx():
    with torch.cuda.stream(comm):
        pass


print("host: with stream(comm)        %.1f us" % host_only(ctx))
print("host: current_stream()         %.1f us" % host_only(lambda: torch.cuda.current_stream(dev)))
v1, v2 = gath[:, :DIM], gath[:, DIM:]
print("host: slicing 2 views          %.1f us" % host_only(lambda: (gath[:n, :DIM], gath[:n, DIM:])))
print("host+gpu: score_emb sp_ (k=300) host %.1f us total %.1f us" % timeit(lambda: engine.score_emb("complex", v1, pe, ent, "sp_")))
print("host+gpu: score_sp (k=300)      host %.1f us total %.1f us" % timeit(lambda: engine.score_sp(engine.Tables("complex", ent, rel), s, p)))
td.destroy_process_group()
