#!/usr/bin/env python3
"""Group launches (kge_score_queries_multi, 8 two-sided batches each) issued on ONE stream against the same launches
alternating between TWO streams (each lane its own query fragments and score buffer): does the next persistent launch
fill the tail / drain / cold start of the one before?  FB15k-237 shape, both query modes; us per group launch."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
E, R, D, n, L = 14541, 237, 512, 512, 8
g = torch.Generator().manual_seed(0)
ent = torch.empty(E, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
rel = torch.empty(R, D).normal_(0, 0.1, generator=g).bfloat16().to(dev)
T = engine.Tables("complex", ent, rel)
pitch, bstride = engine.score_pitch_group(n, E, "sp_po")
for mode, flags in (("single-pass", None), ("split", engine.FLAG_SPLIT_QUERY)):
    lanes = []
    for lane in range(2):
        tri = torch.stack([torch.randint(hi, (L * n,), generator=g) for hi in (E, R, E)], 1).to(dev)
        q = engine.build_queries_group(T, "sp_po", tri, n, L, flags=flags)
        out = torch.empty(L, n, 2, pitch // 2, device=dev)[:, :, :, :E]
        lanes.append((q, out))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    handles = [s.cuda_stream for s in streams]
    res = {}
    for name, pick in (("one_stream", lambda k: 0), ("two_streams", lambda k: k % 2)):
        best = None
        for rep in range(4):
            K = 16
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for st in streams:
                st.wait_event(e0)
            for k in range(K):
                q, out = lanes[k % 2]
                engine.score_queries_group(T, q, out, stream=handles[pick(k)])
            cur = torch.cuda.current_stream(dev)
            for st in streams:
                ev = torch.cuda.Event()
                ev.record(st)
                cur.wait_event(ev)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / K * 1e3
            best = us if best is None else min(best, us)
        res[name] = round(best, 1)
    # same bits whichever way the launches were issued
    ref = [o.clone() for _, o in lanes]
    for lane, (q, out) in enumerate(lanes):
        out.zero_()
        engine.score_queries_group(T, q, out)
    torch.cuda.synchronize()
    same = all(torch.equal(a, o) for a, (_, o) in zip(ref, lanes))
    print(json.dumps({"mode": mode, "us_per_group_launch": res, "us_per_batch": {k: round(v / L, 2) for k, v in res.items()},
                      "bits_equal": same}))
