#!/usr/bin/env python3
"""GraphedStep against the eager loop over two 'epochs' of 100 fixed-shape batches (the FB15k-237 shape), with what a
LibKGE epoch boundary does in between (reseed, a pause, zero_grad(set_to_none)): per-step losses side by side."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import model as km, optim as kopt  # noqa: E402
from kge_amd.train_graph import GraphedStep  # noqa: E402

E, R, D, N = 14541, 237, 512, 512
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(5)
batches = [torch.stack([torch.randint(hi, (N,), generator=g) for hi in (E, R, E)], 1) for _ in range(200)]
BOUNDARY = os.environ.get("BOUNDARY", "seed,sleep,zero").split(",")


def run(graphed):
    torch.manual_seed(0)
    m = km.create("complex", E, R, D, device=dev, score_dtype=torch.bfloat16)
    opt = kopt.Adagrad(m.parameters(), lr=0.1, bf16_copies=True)
    step = GraphedStep(lambda s, p, o, inv: m.loss_sp_po(s, p, o).sum() * inv, opt, warmup=2, enabled=graphed)
    out = []
    out_hold = []
    for k, b in enumerate(batches):
        if k == 100:
            if "seed" in BOUNDARY:
                torch.manual_seed(29)
            if "sleep" in BOUNDARY:
                torch.cuda.synchronize()
                time.sleep(0.5)
            if "train" in BOUNDARY:
                m.train()
            if "hold" in BOUNDARY:   # a different allocator state for the second half
                out_hold.extend(torch.zeros(sz, dtype=torch.uint8, device=dev)
                                for sz in (512, 4096, 12288, 65536, 1 << 20) for _ in range(200))
        if "zero" in BOUNDARY:
            opt.zero_grad(set_to_none=True)          # what TrainingJob.run_epoch does in front of every batch
        bd = b.to(dev)
        inv = torch.full((), 1.0 / N, device=dev)
        out.append(float(step(bd[:, 0], bd[:, 1], bd[:, 2], inv)))
    return out, step


le, _ = run(False)
lg, st = run(True)
bad = [(k, a, b) for k, (a, b) in enumerate(zip(le, lg)) if abs(a - b) > 1e-5 * abs(a)]
dmax = max(abs(a - b) / abs(a) for a, b in zip(le[100:], lg[100:]))
print(f"largest relative difference of a step's loss in the second half: {dmax:.2e}")
print(f"boundary {BOUNDARY}: replays {st.replays}, captures {st.captures}; steps whose loss differs by more than 1e-5: {len(bad)}; "
      f"first {bad[:3]}; mean loss epoch 2: eager {sum(le[100:]) / 100:.5f} graph {sum(lg[100:]) / 100:.5f}")
