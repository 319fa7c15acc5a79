"""d = 128 bf16 ComplEx score_sp / score_sp_po through the one-call entry (route 4 of api.hip's bf16_store_dispatch:
pairs_bf16_v3_kernel) at the FB15k-237 and a Wikidata5M-shard shape, beside d = 256 / 512: us per call, fraction of the
HBM roofline on the algorithmic bytes."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from kge_amd import engine
dev = torch.device("cuda", 0)
def timeit(fn, k=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / k
for E in (14541, 574311):
    for d in (128, 256, 512):
        g = torch.Generator().manual_seed(0)
        ent = (torch.randn(E, d, generator=g) * 0.1).bfloat16().to(dev)
        rel = (torch.randn(237, d, generator=g) * 0.1).bfloat16().to(dev)
        for n in (128, 512, 2048):
            if E > 100000 and n > 512: continue
            s, p, o = (torch.randint(hi, (n,), generator=g).to(dev) for hi in (E, 237, E))
            for split in (0, 1):
                T = engine.Tables("complex", ent, rel, flags=engine.FLAG_SPLIT_QUERY if split else 0)
                for name, fn, sides in (("score_sp", lambda: engine.score_sp(T, s, p, padded=True), 1),
                                        ("score_sp_po", lambda: engine.score_sp_po(T, s, p, o), 2)):
                    us = timeit(fn, 50 if E > 100000 else 200)
                    ab = E * d * 2 + sides * (n * 2 * d * 2 + n * E * 4)
                    print(f"E={E:7d} d={d:3d} n={n:4d} {'split ' if split else 'single'} {name:11s} {us:9.1f} us  frac {ab / (us * 1e-6) / 8e12:.3f}", flush=True)
        del ent, rel
        torch.cuda.empty_cache()
