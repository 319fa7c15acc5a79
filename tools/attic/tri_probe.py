import sys, torch
sys.path.insert(0, ".")
from kge_amd import engine
E_, R, d, n = 14541, 237, 512, 512
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
ent = torch.randn(E_, d, generator=g).to(dev).bfloat16(); rel = torch.randn(2 * R, d, generator=g).to(dev).bfloat16()
T = engine.Tables("complex", ent, rel)
def batch(seed):
    q = torch.Generator().manual_seed(seed)
    return tuple(torch.randint(hi, (n,), generator=q).to(dev) for hi in (E_, R, E_))
b = [batch(1), batch(2)]
tri = [torch.stack(x, 1).contiguous() for x in b]
P = engine.score_pitch(E_)
def ev(fn, K=400):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(K): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3
for comb in ("sp_", "sp_", "sp_po"):
    w = 2 if comb == "sp_po" else 1
    out = torch.empty(n, w * P, device=dev)
    o2 = out.view(n, 2, P)[:, :, :E_] if w == 2 else out[:, :E_]
    for name, nb in (("tuple_noo", [(x[0], x[1], None) for x in b]), ("triples", tri), ("tuple", b), ("tuple_noo", [(x[0], x[1], None) for x in b])):
        if comb == "sp_po" and name == "tuple_noo": continue
        pipe = engine.ScorePipeline(T, comb, n)
        pipe.start(*b[0])
        k = [0]
        def f():
            k[0] += 1
            pipe.step(next_batch=nb[k[0] & 1], out=o2)
        print(comb, name, round(ev(f), 2), "us", flush=True)
