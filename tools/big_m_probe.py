#!/usr/bin/env python3
"""One Wikidata5M/8 shard (574,311 rows, d=256, bf16), n=512: score_sp per launch with the score
rows contiguous (pitch 574,311 floats: rows start at 4-byte granularity) and with the pitch rounded up
to 32 floats (engine.Tables(pad_pitch=True)); and a plain fill / copy of the same-size matrix."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kge_amd import engine  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
E, R, d, n = 574311, 822, 256, 512
g = torch.Generator(device=dev).manual_seed(0)
ent = torch.empty(E, d, device=dev, dtype=torch.bfloat16).normal_(0, 0.1, generator=g)
rel = torch.empty(R, d, device=dev, dtype=torch.bfloat16).normal_(0, 0.1, generator=g)
s = torch.randint(E, (n,), device=dev)
p = torch.randint(R, (n,), device=dev)
ab = bench.algorithmic_bytes(n, E, d)
for pad in (False, True):
    T = engine.Tables("complex", ent, rel, pad_pitch=pad)
    for _ in range(3):
        engine.score_sp(T, s, p)
    ms = bench.event_avg_ms(lambda: engine.score_sp(T, s, p), 20)
    print(f"score_sp one shard pad_pitch={pad}: {ms * 1e3:.1f} us per launch, {ab / ms / 1e6:.0f} GB/s algorithmic "
          f"({ab / ms / 1e6 / 8000:.3f} of 8 TB/s)")
x = torch.empty(n, E, device=dev)
y = torch.empty(n, E, device=dev)
ms = bench.event_avg_ms(lambda: x.fill_(1.0), 20)
print(f"fill_ of the [512, 574311] f32 matrix: {ms * 1e3:.1f} us, {x.numel() * 4 / ms / 1e6:.0f} GB/s written")
ms = bench.event_avg_ms(lambda: y.copy_(x), 20)
print(f"copy_ of it: {ms * 1e3:.1f} us, {x.numel() * 4 / ms / 1e6:.0f} GB/s written (+ as much read)")
