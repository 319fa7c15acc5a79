"""One-call entry at the FB15k-237 shape with the persistent kernel also taking SINGLE batches (switch V8 = 1) against
the default (single batches on pairs_bf16_v7 / v6_kernel): both query modes, one- and two-sided, n = 128 ... 1000."""
import sys, torch
sys.path.insert(0, "/root/repo")
from kge_amd import engine, _lib
dev = torch.device("cuda", 0)
E, R, d = 14541, 237, 512
g = torch.Generator().manual_seed(0)
ent = (torch.randn(E, d, generator=g) * 0.1).bfloat16().to(dev)
rel = (torch.randn(R, d, generator=g) * 0.1).bfloat16().to(dev)
def ev(fn, k=100):
    for _ in range(10): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / k
for n in (128, 256, 512, 768, 1000):
    s, p, o = (torch.randint(hi, (n,), generator=g).to(dev) for hi in (E, R, E))
    for split in (0, 1):
        T = engine.Tables("complex", ent, rel, flags=engine.FLAG_SPLIT_QUERY if split else 0)
        row = []
        for v8 in (None, 1):
            _lib.set_switch("V8", v8)
            if split == 0: _lib.set_switch("ONE_CALL_PREPARED", 1 if v8 else None)  # (route 2 is where single batches meet v8)
            row.append((ev(lambda: engine.score_sp(T, s, p, padded=True)), ev(lambda: engine.score_sp_po(T, s, p, o))))
        _lib.set_switch("V8", None); _lib.set_switch("ONE_CALL_PREPARED", None)
        print(f"n={n:4d} {'split ' if split else 'single'} score_sp {row[0][0]:6.1f} -> {row[1][0]:6.1f} us   score_sp_po {row[0][1]:6.1f} -> {row[1][1]:6.1f} us", flush=True)
