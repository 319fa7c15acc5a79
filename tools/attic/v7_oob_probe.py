"""v7 stores rows straight from the accumulators and relies on the buffer descriptor's range to drop the padded query
rows (>= n) and on an out-of-range per-lane offset to drop columns >= m.  Score into the middle of a NaN-filled
buffer and check that nothing outside [n, m] was written, for ragged n and m."""
import sys
import torch
sys.path.insert(0, ".")
from kge_amd import engine as eng
dev = "cuda:0"
g = torch.Generator().manual_seed(1)
for E, n in ((2111, 63), (2111, 1), (2111, 129), (14541, 511), (4099, 300)):
    ent = (torch.randn(E, 512, generator=g) * 0.3).bfloat16().to(dev)
    rel = (torch.randn(7, 512, generator=g) * 0.3).bfloat16().to(dev)
    T = eng.Tables("complex", ent, rel)
    s, p, o = (torch.randint(hi, (n,), generator=g).to(dev) for hi in (E, 7, E))
    for combine, width in (("sp_", E), ("sp_po", 2 * E)):
        pad_rows, ld = 40, width + 13
        big = torch.full((n + 2 * pad_rows, ld), float("nan"), device=dev)
        out = big[pad_rows:pad_rows + n, 5:5 + width]
        q = eng.build_queries(T, combine, s, p, o if combine == "sp_po" else None)
        eng.score_queries(T, q, out=out)
        torch.cuda.synchronize()
        want = eng.score_sp(T, s, p) if combine == "sp_" else eng.score_sp_po(T, s, p, o)
        ok_in = torch.equal(out, want)
        mask = torch.ones_like(big, dtype=torch.bool)
        mask[pad_rows:pad_rows + n, 5:5 + width] = False
        stray = int((~torch.isnan(big[mask])).sum())
        print(f"E={E} n={n} {combine}: inside equal={ok_in}, stray writes outside={stray}", flush=True)
        assert ok_in and stray == 0
print("ok")
