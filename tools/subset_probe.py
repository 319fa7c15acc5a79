import sys, time, torch
sys.path.insert(0, "/root/repo")
from kge_amd import engine
dev = torch.device("cuda", 0)
E, R, d, n = 14541, 237, 512, 512
g = torch.Generator().manual_seed(0)
ent = torch.empty(E, d).normal_(0, 0.1, generator=g).bfloat16().to(dev)
rel = torch.empty(R, d).normal_(0, 0.1, generator=g).bfloat16().to(dev)
s = torch.randint(E, (n,), generator=g).to(dev); p = torch.randint(R, (n,), generator=g).to(dev)
T = engine.Tables("complex", ent, rel)
def timeit(fn, k=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / k
print("all entities          %.1f us" % timeit(lambda: engine.score_sp(T, s, p)))
for m in (14541, 8192, 2048, 512):
    sub = torch.randperm(E, generator=g)[:m].to(dev)
    print(f"random subset m={m:5d} i64 %.1f us   i32 %.1f us" % (timeit(lambda: engine.score_sp(T, s, p, sub)), timeit(lambda: engine.score_sp(T, s, p, sub.int()))))
    srt = sub.sort().values
    print(f"sorted subset m={m:5d} i64 %.1f us" % timeit(lambda: engine.score_sp(T, s, p, srt)))
