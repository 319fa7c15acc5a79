#!/usr/bin/env python3
"""Cycle stamps of pairs_bf16_v8_rank_kernel: unit periods at the FB15k-237 shape and on a Wikidata5M shard."""
import ctypes
import os
import statistics
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import _lib, engine  # noqa: E402

if os.environ.get("RANK8_LIB"):  # a library built with -DKGE_V8_PROBES (tools/README.md)
    _lib.LIB_PATH = os.path.abspath(os.environ["RANK8_LIB"])
    os.environ["KGE_AMD_BINDING"] = "ctypes"

dev = torch.device("cuda", 0)
# RANK8_PROBES: comma list of KGE_V8R_PROBE values (a library built with -DKGE_V8_PROBES); 0 = the product kernel
PROBES = [int(x) for x in os.environ.get("RANK8_PROBES", "0").split(",")]
KS = [int(x) for x in os.environ.get("RANK8_NFILT", "0,2").split(",")]
SPLIT = os.environ.get("RANK8_SPLIT", "1") == "1"
SHAPES = os.environ.get("RANK8_SHAPES", "fb15k-237,wikidata5m_shard").split(",")


def main():
    L_ = _lib.lib()
    L_.kge_debug_v6_stamps.restype = None
    L_.kge_debug_v6_stamps.argtypes = [ctypes.c_void_p]
    rng = np.random.default_rng(0)
    n = 512
    for tag, E, R, d in (("fb15k-237", 14541, 237, 512), ("wikidata5m_shard", (4594485 + 7) // 8, 822, 256)):
        if tag not in SHAPES:
            continue
        g = torch.Generator(device=dev).manual_seed(7)
        ent = (torch.randn(E, d, generator=g, device=dev) * 0.3).bfloat16()
        rel = (torch.randn(R, d, generator=g, device=dev) * 0.3).bfloat16()
        s, p, o = (torch.from_numpy(rng.integers(0, hi, n)).to(dev) for hi in (E, R, E))
        for K in KS:
            lists = []
            for tc in (o.cpu().numpy(), s.cpu().numpy()):
                per = [np.unique(np.append(rng.integers(0, E, 4), c)) for c in tc]
                end = np.cumsum([len(x) for x in per])
                beg = end - np.array([len(x) for x in per])
                one = tuple(torch.from_numpy(np.asarray(x, np.int64)).to(dev) for x in (beg, end, np.concatenate(per)))
                lists.append([one] * K)
            for flags, probe in [(0, pr) for pr in PROBES] + ([(engine.FLAG_SPLIT_QUERY, 0)] if SPLIT else []):
                os.environ["KGE_V8R_PROBE"] = str(probe)
                T = engine.Tables("complex", ent, rel, flags=flags)
                t_sp = engine.score_sp(T, s, p, o).diagonal().contiguous()
                t_po = engine.score_po(T, p, o, s).diagonal().contiguous()
                cnt = torch.zeros(2, 2, K + 1, n, dtype=torch.int64, device=dev)

                def fused():
                    assert engine.score_rank_sp_po(T, s, p, o, t_sp, t_po, lists[0], lists[1], 1e-5, 1e-4, cnt[0, 0],
                                                   cnt[0, 1], cnt[1, 0], cnt[1, 1])
                for _ in range(3):
                    fused()
                st = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
                torch.cuda.synchronize()
                L_.kge_debug_v6_stamps(ctypes.c_void_p(st.data_ptr()))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fused()
                e1.record()
                torch.cuda.synchronize()
                L_.kge_debug_v6_stamps(None)
                v = st.view(4096, 64).cpu()
                v = v[(v[:, 0] != 0) & (v[:, 2] != 0)]
                nst = int((v[0, :32] != 0).sum())
                rel_ = (v[:, :nst] - v[:, :1]).double().median(dim=0).values
                per = [float(rel_[i + 1] - rel_[i]) for i in range(2, nst - 1)]
                whole = ((v[:, 40] - v[:, 1]).double() / v[:, 41].double().clamp(min=1))
                tot = (v[:, 40] - v[:, 0]).double()
                real_us = float(v[:, 43].max() - v[:, 42].min()) / 100.0
                print(f"    kernel by the 100 MHz real-time counter (first workgroup start -> last workgroup end): {real_us:.1f} us "
                      f"of the {e0.elapsed_time(e1) * 1e3:.1f} us call -> shader clock {float(tot.max()) / real_us / 1e3:.2f} GHz")
                print(f"    whole run: cycles per unit (end - R0) / units: median {float(whole.median()):.0f}  "
                      f"max {float(whole.max()):.0f}; units per workgroup {int(v[:, 41].median())}; workgroup cycles "
                      f"median {float(tot.median()):.0f} max {float(tot.max()):.0f} -> clock "
                      f"{float(tot.max()) / (e0.elapsed_time(e1) * 1e3) / 1e3:.2f} GHz if the call were this kernel alone")
                print(f"{tag} filters={K} split={int(bool(flags))} probe={os.environ.get('KGE_V8R_PROBE', '0')}: {v.shape[0]} workgroups; R0 {float(rel_[1]):.0f}; unit periods "
                      f"{[round(x) for x in per[:10]]} median {statistics.median(per) if per else 0:.0f}; call {e0.elapsed_time(e1) * 1e3:.1f} us",
                      flush=True)
        del ent, rel
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
