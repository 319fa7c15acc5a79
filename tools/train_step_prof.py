#!/usr/bin/env python3
"""The 1vsAll training step of bench.py's roofline_train leg (bf16 scoring copies, Adagrad), STEPS times: run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split (tools/gpu_trainprof.sh), or alone for wall clock and the
host's share (time to ISSUE a step vs time to finish it)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import model as km, optim as kopt, _lib  # noqa: E402

# KGE_SWITCHES="BWD_FORK=0,CE_V8=0": the library's measurement switches (kge_amd/csrc/switches.hpp) for A/B runs
for kv in filter(None, os.environ.get("KGE_SWITCHES", "").split(",")):
    k, v = kv.split("=")
    _lib.set_switch(k, int(v))

E, R, D, N = 14541, 237, 512, 512
STEPS = int(os.environ.get("STEPS", "200"))
dev = torch.device("cuda", 0)
q = torch.Generator().manual_seed(3)
s, p, o = (torch.randint(hi, (N,), generator=q).to(dev) for hi in (E, R, E))
for tag, sd in (("bf16_scoring", torch.bfloat16), ("f32_scoring", torch.float32)):
    if os.environ.get("ONLY") and os.environ["ONLY"] != tag:
        continue
    torch.manual_seed(0)
    m = km.create("complex", E, R, D, device=dev, score_dtype=sd)
    opt = kopt.Adagrad(m.parameters(), lr=0.1, bf16_copies=(sd == torch.bfloat16))

    def step():
        opt.zero_grad(set_to_none=True)
        m.loss_sp_po_sum(s, p, o).backward()
        opt.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{tag}: {1e3 * (t2 - t0) / STEPS:.4f} ms per step wall clock; host issue {1e3 * (t1 - t0) / STEPS:.4f} ms per step "
          f"(the GPU was {'behind' if t2 - t1 > 0.05 * (t1 - t0) else 'waiting for'} the host)", flush=True)
    if os.environ.get("GRAPH", "1") == "1":
        # the same step as kge_amd.train_graph.GraphedStep replays it (static index buffers and root gradient; forward,
        # backward and optimizer in one hipGraph)
        from kge_amd.train_graph import GraphedStep
        tri = torch.stack([s, p, o], 1)
        gs = GraphedStep(lambda t_: m.loss_sp_po_sum(t_[:, 0], t_[:, 1], t_[:, 2]), opt, warmup=1)
        for _ in range(3):
            gs(tri)
        assert gs.replays > 0, gs.disabled_reason
        if os.environ.get("IN_PLACE", "1") == "1":  # the batch written into the captured step's own input buffer
            gs.static_inputs[0].copy_(tri)
            tri = gs.static_inputs[0]
        for _ in range(5):
            gs(tri)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            gs(tri)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"{tag}: {1e3 * (time.perf_counter() - t0) / STEPS:.4f} ms per step as ONE hipGraph replay "
              f"(host issue {1e3 * (t1 - t0) / STEPS:.4f} ms per step)", flush=True)
