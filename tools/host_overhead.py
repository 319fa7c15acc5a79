import time, torch, sys
sys.path.insert(0, "/root/repo")
from kge_amd import engine
import bench
dev = torch.device("cuda", 0)
ent, rel, s, p, o = bench.make_inputs(0, dev, 512)
T = engine.Tables("complex", ent, rel)
for _ in range(50): engine.score_sp(T, s, p)
torch.cuda.synchronize()
for K in (200, 1000):
    t0 = time.perf_counter()
    for _ in range(K): engine.score_sp(T, s, p)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K}: host issue {1e6*(t1-t0)/K:.1f} us/call, total {1e6*(t2-t0)/K:.1f} us/call")
out = torch.empty(512, 14541, device=dev)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): engine.score_sp(T, s, p)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
