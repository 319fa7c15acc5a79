#!/usr/bin/env python3
"""How much of a ranking batch would band-and-rescore leave undecided?  

Band-and-rescore counts with the single-pass score x1 = q_hi . t and decides a (row, column) pair only if
|x1 - true| clears the reference's tie band (eval_entity_ranking.py:571-596: isclose with rtol 1e-4, atol 1e-5 -- the
job's defaults) PLUS a bound on what the single pass left out, lo = q_lo . t, |lo| <= ||q_lo|| ||t|| (Cauchy-Schwarz with
the row's own q_lo: the tightest bound that needs no second pass).  Undecided pairs need the second chain.  This tool
measures, on tables like the bench's and the rank fixtures' (normal entries) at the Wikidata5M-shard shape:
  * the share of undecided pairs,
  * the share of 32-row x 64-column tiles of the counting kernel that hold at least one (a tile with one needs the
    q_lo chain for the whole tile),
with the true object (a) drawn at random, as the bench and the fixtures draw it, and (b) in the top tail of its row (the
column at the 99.99th percentile of the row's scores: what a trained model's true scores look like)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = torch.device("cuda", 0)
E, R, d, n = int(os.environ.get("E", 574311)), 822, int(os.environ.get("D", 256)), 512
RTOL, ATOL = 1e-4, 1e-5
g = torch.Generator().manual_seed(5)
for scale in (0.1, 0.5):
    ent = (torch.randn(E, d, generator=g) * scale).to(torch.bfloat16).to(dev)
    rel = (torch.randn(R, d, generator=g) * scale).to(torch.bfloat16).to(dev)
    s = torch.randint(E, (n,), generator=g).to(dev)
    p = torch.randint(R, (n,), generator=g).to(dev)
    h = d // 2
    a, r = ent[s].float(), rel[p].float()
    # ComplEx sp_ query: q = s (x) r (complex product), scores = Re<q, conj(t)> = q_re . t_re + q_im . t_im
    q = torch.cat([a[:, :h] * r[:, :h] - a[:, h:] * r[:, h:], a[:, h:] * r[:, :h] + a[:, :h] * r[:, h:]], 1)
    q_hi = q.to(torch.bfloat16).float()
    q_lo = (q - q_hi).to(torch.bfloat16).float()
    T = ent.float()
    x1 = q_hi @ T.t()                       # the single-pass score
    lo = q_lo @ T.t()                       # what the second chain adds
    bound = q_lo.norm(dim=1, keepdim=True) * T.norm(dim=1).view(1, -1)
    print(f"tables N(0, {scale}^2), ComplEx d = {d}, {n} sp_ queries x {E} entities", flush=True)
    print(f"  |lo| / bound: mean {float((lo.abs() / bound).mean()):.3f}, max {float((lo.abs() / bound).max()):.3f}   "
          f"||q_lo|| / ||q||: {float((q_lo.norm(dim=1) / q.norm(dim=1)).mean()):.2e}")
    sigma = (x1.std(dim=1)).mean()
    for name in ("true object at random", "true object at the row's 99.99th percentile"):
        if name.startswith("true object at random"):
            o = torch.randint(E, (n,), generator=g).to(dev)
        else:
            k = max(1, int(E * 1e-4))
            o = torch.topk(x1, k, dim=1).indices[:, -1]
        true = (x1 + lo).gather(1, o.view(-1, 1))
        band = ATOL + RTOL * true.abs()
        und = (x1 - true).abs() <= band + bound
        rows32 = und.view(n // 32, 32, E)
        pad = (-E) % 64
        tiles = torch.nn.functional.pad(rows32.any(dim=1), (0, pad)).view(n // 32, (E + pad) // 64, 64).any(dim=2)
        wrong = ((x1 > true) != ((x1 + lo) > true)) & ~und  # decided pairs the single pass would decide WRONGLY: must be 0
        print(f"  {name}: true score at {float(((true.view(-1) - x1.mean(dim=1)) / x1.std(dim=1)).mean()):+.2f} sigma; "
              f"undecided pairs {float(und.float().mean()):.3e} ({int(und.sum())} of {n * E}), "
              f"tiles (32 x 64) holding one {float(tiles.float().mean()):.4f}; decided-but-wrong {int(wrong.sum())}")
    print(f"  (band half-width / sigma of a row's scores: {float((bound.mean() + ATOL) / sigma):.4f})", flush=True)
    del x1, lo, bound, T
    torch.cuda.empty_cache()
