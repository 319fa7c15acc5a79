import sys, torch
sys.path.insert(0, "/root/repo")
from kge_amd import engine
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
E, R, d, n = 14541, 237, 512, 512
g = torch.Generator().manual_seed(0)
for name in ("transe", "rotate"):
    dr = d // 2 if name == "rotate" else d
    ent = torch.empty(E, d).normal_(0, 0.1, generator=g).to(dev)
    rel = torch.empty(R, dr).normal_(0, 0.1, generator=g).to(dev)
    s = torch.randint(E, (n,), generator=g).to(dev); p = torch.randint(R, (n,), generator=g).to(dev)
    T = engine.Tables(name, ent, rel)
    gout = torch.randn(n, E, device=dev)
    sc = engine.score_sp(T, s, p)
    for _ in range(2): engine.score_pairs_bwd(T, "sp", s, p, None, gout, sc)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            engine.score_sp(T, s, p); engine.score_pairs_bwd(T, "sp", s, p, None, gout, sc)
        torch.cuda.synchronize()
    for e in prof.key_averages():
        if "kge::" in e.key:
            print(name, e.key[:70], "%.1f us" % (e.device_time_total / e.count))
