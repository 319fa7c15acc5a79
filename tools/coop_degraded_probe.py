"""tools/d128_probe.py's sequence with the workspace's DEGRADED word (the countdown a timed-out cooperative query build
leaves: pairs_bf16_v4_kernel / pairs_bf16_v3_kernel, flags + 512 * 8) read after every configuration."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from kge_amd import engine
dev = torch.device("cuda", 0)
def degraded():
    torch.cuda.synchronize()
    out = []
    for k, b in engine._WORKSPACES.items():
        out.append(int(b[32768:32776].view(torch.int64)[0]))
    return out
def timeit(fn, k):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / k
E = 14541
for d in (128, 256, 512):
    g = torch.Generator().manual_seed(0)
    ent = (torch.randn(E, d, generator=g) * 0.1).bfloat16().to(dev)
    rel = (torch.randn(237, d, generator=g) * 0.1).bfloat16().to(dev)
    for n in (128, 512, 2048):
        s, p, o = (torch.randint(hi, (n,), generator=g).to(dev) for hi in (E, 237, E))
        for split in (0, 1):
            T = engine.Tables("complex", ent, rel, flags=engine.FLAG_SPLIT_QUERY if split else 0)
            for name, fn in (("score_sp", lambda: engine.score_sp(T, s, p, padded=True)), ("score_sp_po", lambda: engine.score_sp_po(T, s, p, o))):
                us = timeit(fn, 200)
                print(f"d={d:3d} n={n:4d} {'split ' if split else 'single'} {name:11s} {us:9.1f} us   degraded words {degraded()}", flush=True)
    del ent, rel
    torch.cuda.empty_cache()
