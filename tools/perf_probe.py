#!/usr/bin/env python3
"""Per-kernel timing probe on one MI355X (HIP events on the launch stream, interleaved rounds).
Writes one JSON line per measurement; used to fill DESIGN.md's per-kernel roofline table."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    # back-to-back (launch overhead pipelined, as in bench.py): one event pair around `iters` calls
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return t[len(t) // 2], t[0], a.elapsed_time(b) * 1e3 / iters


def tables(model, E, R, d, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    ent = torch.empty(E, d).normal_(0, 0.1, generator=g)
    dr = d // 2 if model == "rotate" else d
    rel = torch.empty(R, dr).uniform_(-3.14, 3.14, generator=g) if model == "rotate" else torch.empty(R, dr).normal_(0, 0.1, generator=g)
    return engine.Tables(model, ent.to(dtype).to(dev), rel.to(dtype).to(dev))


def emit(**kw):
    print(json.dumps(kw), flush=True)


def main():
    q = torch.Generator().manual_seed(1)
    # ---- reference points: plain HBM write / copy of the score-matrix size (torch kernels)
    x = torch.empty(512, 14541, device=dev)
    y = torch.empty_like(x)
    med, mn, avg = timeit(lambda: x.fill_(1.0))
    emit(kernel="torch.fill_ 29.8MB", us_med=med, us_min=mn, gbs=x.numel() * 4 / med / 1e3)
    med, mn, avg = timeit(lambda: y.copy_(x))
    emit(kernel="torch.copy_ 29.8MB", us_med=med, us_min=mn, gbs=2 * x.numel() * 4 / med / 1e3)
    big = torch.empty(64 * 1024 * 1024, device=dev)
    med, mn, avg = timeit(lambda: big.fill_(1.0))
    emit(kernel="torch.fill_ 256MB", us_med=med, us_min=mn, gbs=big.numel() * 4 / med / 1e3)
    del x, y, big
    # ---- pair kernels (sp_) at FB15k-237 shape
    E, R, d = 14541, 237, 512
    for n in (128, 512, 1024):
        s = torch.randint(E, (n,), generator=q).to(dev); p = torch.randint(R, (n,), generator=q).to(dev)
        for model, dtype, flags, tag in [
            ("complex", torch.bfloat16, 0, "bf16-mfma"),
            ("complex", torch.bfloat16, engine.FLAG_BF16_V3, "bf16-mfma-v3-coop"),
            ("complex", torch.bfloat16, -1, "bf16-mfma-v3-noworkspace"),
            ("complex", torch.bfloat16, engine.FLAG_BF16_V1, "bf16-mfma-v1"),
            ("distmult", torch.bfloat16, 0, "bf16-mfma"),
            ("complex", torch.bfloat16, engine.FLAG_EXACT, "bf16-exact-f32mfma"),
            ("complex", torch.float32, 0, "f32-mfma"),
            ("complex", torch.float32, engine.FLAG_NO_MFMA, "f32-valu"),
            ("transe", torch.float32, 0, "f32-valu"),
            ("rotate", torch.float32, 0, "f32-valu"),
        ]:
            if n != 512 and tag not in ("bf16-mfma", "bf16-mfma-v3-coop"):
                continue
            T = tables(model, E, R, d, dtype)
            if flags == -1:
                T.use_workspace, flags = False, 0
            med, mn, avg = timeit(lambda: engine.score_sp(T, s, p, flags=flags))
            elt = 2 if dtype == torch.bfloat16 else 4
            byts = E * d * elt + n * 2 * d * elt + n * E * 4
            emit(kernel="score_sp", model=model, tag=tag, n=n, E=E, d=d, us_pipelined=avg, us_med=med, us_min=mn,
                 gbs=byts / med / 1e3, gflops=2.0 * n * E * d / med / 1e3, triples_per_s=n * E / med * 1e6)
    # ---- score_sp_po (EntityRankingJob's call): one two-sided launch vs two one-sided calls
    for n in (128, 512):
        s = torch.randint(E, (n,), generator=q).to(dev); p = torch.randint(R, (n,), generator=q).to(dev)
        o = torch.randint(E, (n,), generator=q).to(dev)
        T = tables("complex", E, R, d, torch.bfloat16)
        byts = 2 * (E * d * 2 + n * 2 * d * 2 + n * E * 4)
        med, mn, avg = timeit(lambda: engine.score_sp_po(T, s, p, o))
        emit(kernel="score_sp_po", tag="two-sided launch", n=n, us_pipelined=avg, us_med=med, us_min=mn, gbs=byts / med / 1e3,
             triples_per_s=2 * n * E / med * 1e6)
        med, mn, avg = timeit(lambda: engine.score_sp_po(T, s, p, o, flags=engine.FLAG_BF16_V3))
        emit(kernel="score_sp_po", tag="two v3 launches", n=n, us_pipelined=avg, us_med=med, us_min=mn, gbs=byts / med / 1e3,
             triples_per_s=2 * n * E / med * 1e6)
        med, mn, avg = timeit(lambda: (engine.score_sp(T, s, p), engine.score_po(T, p, o)))
        emit(kernel="score_sp+score_po", tag="two v4 launches", n=n, us_pipelined=avg, us_med=med, us_min=mn, gbs=byts / med / 1e3,
             triples_per_s=2 * n * E / med * 1e6)
    # ---- spo / negatives at WN18RR shape (gather bound)
    E, R, d = 40943, 11, 512
    for model in ("rotate", "transe", "complex", "distmult"):
        for dtype in (torch.float32, torch.bfloat16):
            T = tables(model, E, R, d, dtype)
            n, K = 512, 1000
            s = torch.randint(E, (n,), generator=q).to(dev); p = torch.randint(R, (n,), generator=q).to(dev)
            o = torch.randint(E, (n,), generator=q).to(dev)
            neg = torch.randint(E, (n, K), generator=q).to(dev)
            elt = 2 if dtype == torch.bfloat16 else 4
            for slot in (0, 2):
                med, mn, avg = timeit(lambda: engine.score_neg(T, s, p, o, slot, neg), iters=15)
                byts = n * K * (d * elt + 4 + 8)
                emit(kernel="score_neg", model=model, dtype=str(dtype), slot=slot, n=n, K=K, us_med=med,
                     us_min=mn, gbs=byts / med / 1e3, triples_per_s=n * K / med * 1e6)
            N = n * K
            ss = s.repeat_interleave(K); pp = p.repeat_interleave(K); oo = neg.reshape(-1)
            med, mn, avg = timeit(lambda: engine.score_spo(T, ss, pp, oo), iters=15)
            emit(kernel="score_spo", model=model, dtype=str(dtype), N=N, us_med=med, us_min=mn,
                 gbs=N * (3 * d * elt + 4) / med / 1e3, triples_per_s=N / med * 1e6)
    # ---- rank kernel: HBM scan of the score matrix
    for n, c in ((512, 14541), (512, 2 * 14541), (128, 500000)):
        sc = torch.randn(n, c, device=dev)
        tr = sc[torch.arange(n), torch.randint(c, (n,))].clone()
        med, mn, avg = timeit(lambda: engine.rank_counts(sc, tr))
        emit(kernel="rank_counts", n=n, c=c, us_med=med, us_min=mn, gbs=n * c * 4 / med / 1e3)


if __name__ == "__main__":
    main()
