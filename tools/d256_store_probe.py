"""d = 256 group store (pairs_bf16_v8_ce_kernel<128, V3_STORE>) under the three store cache policies
   python tools/d256_store_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kge_amd import _lib, engine  # noqa: E402

dev = torch.device("cuda:0")
n = bench.BATCH
for tag, E, L in (("fb15k-237 shape", bench.E_FB, 8), ("wikidata5m shard", (bench.E_WD + 7) // 8, 2)):
    g = torch.Generator(device=dev).manual_seed(1)
    ent = (torch.randn(E, 256, generator=g, device=dev) * 0.1).bfloat16()
    rel = (torch.randn(bench.R_FB, 256, generator=g, device=dev) * 0.1).bfloat16()
    T = engine.Tables("complex", ent, rel)
    tri = torch.stack([torch.randint(hi, (n * L,), device=dev) for hi in (E, bench.R_FB, E)], 1)
    q = engine.build_queries_group(T, "sp_", tri, n, L)
    pitch = engine.score_pitch(E)
    out = torch.empty(L, n, pitch, device=dev)
    ab = L * bench.algorithmic_bytes(n, E, 256)
    for pol in (None, 0, 1, 2):
        _lib.set_switch("V4_STORE_SC1", pol)
        for _ in range(3):
            engine.score_queries_group(T, q, out[:, :, :E])
        ms = bench.event_avg_ms(lambda: engine.score_queries_group(T, q, out[:, :, :E]), 10)
        print(f"{tag}: policy {pol}: {ms * 1e3:.1f} us per launch of {L}, {ms * 1e3 / L:.1f} us per batch, frac {ab / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS:.3f}")
    _lib.set_switch("V4_STORE_SC1", None)
    # one launch per batch on the single-batch kernels (what a d = 256 group was until round 6)
    _lib.set_switch("V8", 0)
    qs = [engine.build_queries(T, "sp_", tri[l * n:(l + 1) * n, 0], tri[l * n:(l + 1) * n, 1], None) for l in range(L)]

    def one_by_one():
        for l in range(L):
            engine.score_queries(T, qs[l], out=out[l, :, :E])
    for _ in range(3):
        one_by_one()
    ms = bench.event_avg_ms(one_by_one, 10)
    print(f"{tag}: V8=0 (one launch per batch, round-3 kernels): {ms * 1e3 / L:.1f} us per batch, frac {ab / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS:.3f}")
    _lib.set_switch("V8", None)
    del T, q, qs, out, ent, rel
    torch.cuda.empty_cache()
