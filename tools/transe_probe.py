"""TransE / RotatE score_sp at the FB15k-237 shape: the packed kernel against the generic one (switch TRANSE_GENERIC)
   python tools/transe_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kge_amd import _lib, engine  # noqa: E402

dev = torch.device("cuda:0")
E, R, d, n = bench.E_FB, bench.R_FB, bench.DIM, bench.BATCH
g = torch.Generator(device=dev).manual_seed(3)
for dtype in (torch.float32, torch.bfloat16):
    ent = (torch.randn(E, d, generator=g, device=dev) * 0.3).to(dtype)
    rel = (torch.randn(R, d, generator=g, device=dev) * 0.3).to(dtype)
    s, p = (torch.randint(hi, (n,), device=dev) for hi in (E, R))
    for l_norm in (1.0, 2.0):
        T = engine.Tables("transe", ent, rel, l_norm)
        res = {}
        for which in ("generic", "packed"):
            _lib.set_switch("TRANSE_GENERIC", 1 if which == "generic" else 0)
            for _ in range(3):
                out = engine.score_sp(T, s, p, padded=True)
            res[which] = (bench.event_avg_ms(lambda: engine.score_sp(T, s, p, padded=True), 20) * 1e3, out)
        _lib.set_switch("TRANSE_GENERIC", None)
        lane_ops = 2.0 * n * E * d  # one subtract + one accumulate per element
        peak = 256 * 4 * 16 * 2.4e9
        print(f"transe {dtype} l_norm {l_norm}: generic {res['generic'][0]:.1f} us, packed {res['packed'][0]:.1f} us "
              f"(= {lane_ops / (res['packed'][0] * 1e-6) / peak:.2f} of the VALU issue peak at 2 lane-ops per element), "
              f"equal bits: {torch.equal(res['generic'][1], res['packed'][1])}")
