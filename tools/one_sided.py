#!/usr/bin/env python3
"""K one-sided launches (score_sp, C2 shape) for profiler runs: `--pad` returns a view of a matrix
with the row pitch rounded up to 32 floats (engine.Tables(pad_pitch=True)).  Prints the HIP-event
average per launch."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kge_amd import engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--pad", action="store_true")
ap.add_argument("--n", type=int, default=512)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ent, rel, s, p, o = bench.make_inputs(0, dev, a.n)
T = engine.Tables("complex", ent, rel, pad_pitch=a.pad)
for _ in range(10):
    engine.score_sp(T, s, p)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(a.steps):
    engine.score_sp(T, s, p)
e1.record()
torch.cuda.synchronize()
print(f"one-sided score_sp n={a.n} pad={a.pad}: {e0.elapsed_time(e1) / a.steps * 1e3:.2f} us per launch")
