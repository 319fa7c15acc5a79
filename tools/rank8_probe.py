#!/usr/bin/env python3
"""The counting kernel (kge_score_rank_sp_po) at the FB15k-237 shape (d = 512) and on one of eight Wikidata5M shards
(d = 256): pairs_bf16_v8_rank_kernel (two consumer waves per SIMD), single-pass and split queries, against the
round-3 epilogue of pairs_bf16_v4_kernel (KGE_V8_RANK=0) and the two-step path.  HIP events, us per batch; fraction
of the bf16 matrix peak by ALGORITHMIC flops (2 directions x 2 n E d) and by executed flops (split: twice)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
PEAK = 2500.0
NFILT = [int(x) for x in os.environ.get("RANK8_NFILT", "2").split(",")]


def ev(fn, steps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def main():
    rng = np.random.default_rng(0)
    n = 512
    for tag, E, R, d in (("fb15k-237", 14541, 237, 512), ("wikidata5m_shard", (4594485 + 7) // 8, 822, 256)):
        g = torch.Generator(device=dev).manual_seed(7)
        ent = (torch.randn(E, d, generator=g, device=dev) * 0.3).bfloat16()
        rel = (torch.randn(R, d, generator=g, device=dev) * 0.3).bfloat16()
        s, p, o = (torch.from_numpy(rng.integers(0, hi, n)).to(dev) for hi in (E, R, E))
        lists = []
        for tc in (o.cpu().numpy(), s.cpu().numpy()):
            per = [np.unique(np.append(rng.integers(0, E, 4), c)) for c in tc]
            end = np.cumsum([len(x) for x in per])
            beg = end - np.array([len(x) for x in per])
            one = tuple(torch.from_numpy(np.asarray(x, np.int64)).to(dev) for x in (beg, end, np.concatenate(per)))
            lists.append([one, one])
        flops = 2.0 * 2.0 * n * E * d
        for mode, flags, env in (("v8 single-pass", 0, None), ("v8 split", engine.FLAG_SPLIT_QUERY, None),
                                 ("v4 single-pass (round 3)", 0, "0")):
            T = engine.Tables("complex", ent, rel, flags=flags)
            if env is None:
                os.environ.pop("KGE_V8_RANK", None)
            else:
                os.environ["KGE_V8_RANK"] = env
            t_sp = engine.score_sp(T, s, p, o).diagonal().contiguous()
            t_po = engine.score_po(T, p, o, s).diagonal().contiguous()
            for nf in NFILT:   # filter sets per direction: 0 = raw ranks only (no filter words)
                cnt = torch.zeros(2, 2, nf + 1, n, dtype=torch.int64, device=dev)

                def fused():
                    ok = engine.score_rank_sp_po(T, s, p, o, t_sp, t_po, lists[0][:nf], lists[1][:nf], 1e-5, 1e-4,
                                                 cnt[0, 0], cnt[0, 1], cnt[1, 0], cnt[1, 1])
                    assert ok
                us = ev(fused, 30 if E < 100000 else 8)
                ex = 2.0 if flags else 1.0
                print(json.dumps({"shape": tag, "mode": mode, "filter_sets": nf, "us_per_batch": round(us, 1),
                                  "frac_of_bf16_peak_algorithmic": round(flops / (us * 1e-6) / 1e12 / PEAK, 3),
                                  "frac_of_bf16_peak_executed": round(ex * flops / (us * 1e-6) / 1e12 / PEAK, 3)}),
                      flush=True)
        os.environ.pop("KGE_V8_RANK", None)
        del ent, rel
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
