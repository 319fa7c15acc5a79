#!/usr/bin/env python3
"""Timing of forward + backward of one 1vsAll score_sp call (ComplEx / DistMult / TransE, f32
tables, C2 shape) through kge_amd's autograd glue, next to plain torch ops on the same GPU
(the reference's op sequence written out below, run on the device)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine
from kge_amd import model as km

dev = torch.device("cuda", 0)
E, R, d, n = 14541, 237, 512, 512
g = torch.Generator().manual_seed(0)


def timeit(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / k


for name in ("complex", "distmult", "transe"):
    ent = torch.empty(E, d).normal_(0, 0.1, generator=g).to(dev).requires_grad_(True)
    rel = torch.empty(R, d).normal_(0, 0.1, generator=g).to(dev).requires_grad_(True)
    s = torch.randint(E, (n,), generator=g).to(dev); p = torch.randint(R, (n,), generator=g).to(dev)
    w = torch.randn(n, E, device=dev)

    def ours_fwd():
        return km._ScorePairs.apply(name, 1.0, "sp", ent, rel, s, p, None)

    def ours():
        ent.grad = rel.grad = None
        (ours_fwd() * w).sum().backward()

    def ref_fwd():  # the reference's op sequence (complex.py:24-37, distmult.py:18, transe.py:24-28)
        s_emb, p_emb, o_emb = ent[s], rel[p], ent[torch.arange(E, device=dev)]
        if name == "distmult":
            return (s_emb * p_emb).mm(o_emb.t())
        if name == "transe":
            return -torch.cdist(s_emb + p_emb, o_emb, p=1.0, compute_mode="donot_use_mm_for_euclid_dist")
        p_re, p_im = (t.contiguous() for t in p_emb.chunk(2, dim=1))
        o_re, o_im = (t.contiguous() for t in o_emb.chunk(2, dim=1))
        s_all = torch.cat((s_emb, s_emb), dim=1)
        r_all = torch.cat((p_re, p_emb, -p_im), dim=1)
        o_all = torch.cat((o_emb, o_im, o_re), dim=1)
        return (s_all * r_all).mm(o_all.t())

    def ref():
        ent.grad = rel.grad = None
        (ref_fwd() * w).sum().backward()

    with torch.no_grad():
        tf = timeit(ours_fwd)
    to = timeit(ours)
    try:
        with torch.no_grad():
            rf = timeit(ref_fwd)
        tr = timeit(ref)
    except Exception as e:
        rf = tr = float("nan"); print("torch ops failed:", type(e).__name__, str(e)[:100])
    print(f"{name:9s} f32 n={n}: ours fwd {tf:8.1f} us, fwd+bwd {to:8.1f} us | torch ops on GPU fwd {rf:8.1f} us, fwd+bwd {tr:8.1f} us")

# ---- pieces of the ComplEx backward
name = "complex"
ent = torch.empty(E, d).normal_(0, 0.1, generator=g).to(dev)
rel = torch.empty(R, d).normal_(0, 0.1, generator=g).to(dev)
s = torch.randint(E, (n,), generator=g).to(dev); p = torch.randint(R, (n,), generator=g).to(dev)
T = engine.Tables(name, ent, rel)
gout = torch.randn(n, E, device=dev)
scores = engine.score_sp(T, s, p)
print("score_pairs_bwd (2 GEMMs + build + chain): %.1f us" % timeit(lambda: engine.score_pairs_bwd(T, "sp", s, p, None, gout, scores)))
Te = engine.Tables(name, ent, rel, flags=engine.FLAG_EXACT)
print("score_pairs_bwd, self-contained kernels:   %.1f us" % timeit(lambda: engine.score_pairs_bwd(Te, "sp", s, p, None, gout, scores), k=5))
q = torch.randn(n, d, device=dev)
print("torch.mm  gout[n,E] @ ent[E,d]:            %.1f us" % timeit(lambda: gout @ ent))
print("torch.mm  gout.t()[E,n] @ q[n,d]:          %.1f us" % timeit(lambda: gout.t() @ q))
print("torch.zeros_like(ent):                     %.1f us" % timeit(lambda: torch.zeros_like(ent)))
ge = torch.zeros_like(ent); gt = torch.randn_like(ent)
print("ge += g_t:                                 %.1f us" % timeit(lambda: ge.add_(gt)))
