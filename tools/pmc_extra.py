#!/usr/bin/env python3
"""Summarise tools/gpu_pmc.sh's passes into profiles/<tag>_rocprofv3_pmc_mfma_gather.txt:
MFMA utilisation of the headline kernel (SQ_VALU_MFMA_BUSY_CYCLES against SIMD-cycles of the
kernel's duration) and the HBM-side traffic / bandwidth of the gather-bound negative-sampling kernels.
Usage: python tools/pmc_extra.py gpurun_out/<tag> <tag> [kernel_avg_us_of_v4]"""
import sqlite3
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
v4_us = float(sys.argv[3]) if len(sys.argv) > 3 else 18.2   # profiles/*_rocprofv3_kernel_stats.txt
CLK, CUS, SIMDS = 2.4e9, 256, 4


def counters(sub):
    d = sqlite3.connect(f"{out_dir}/{sub}/r_results.db")
    res = {}
    for k, c, n, v in d.execute(
            "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by 1,2"):
        res[(k.split("(")[0].replace("void ", ""), c)] = (n, v)
    return res


L = [f"{tag}: rocprofv3 --pmc passes of tools/gpu_pmc.sh (each counter set in its own run, no trace domains)", ""]
L.append("== headline kernel (bench.py --steps 20 --warmup 5): matrix-core utilisation")
m = {}
for sub in ("mfma_a", "mfma_b", "mfma_c"):
    for (k, c), (n, v) in counters(sub).items():
        if "pairs_bf16_v4" in k:
            m[c] = v
            L.append(f"  {sub}  {c:30s} dispatches={n:4d} mean/dispatch={v:14.1f}")
busy = m["SQ_VALU_MFMA_BUSY_CYCLES"]
n_mfma = busy / 32.0
sides = int(round(n_mfma / 233472.0))          # 1: score_sp launches, 2: two-sided score_sp_po launches
wgs = 228 if sides == 1 else 232               # workgroups of the launch (one per CU)
flops = n_mfma * 32768.0
L += ["",
      f"  MFMAs per launch = BUSY/32 = {n_mfma:.0f}  (= {sides} x 512 x 14592 x 512 / 16384 = {sides * 233472} v_mfma_f32_32x32x16_bf16:",
      f"    no redundant MFMA work; SQ_INSTS_VALU_MFMA_MOPS_BF16 = {m.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0):.0f} = 64 x that: 512-flop units)",
      f"  kernel duration {v4_us} us (rocprofv3 --kernel-trace, same command) = {v4_us*1e-6*CLK:.0f} cycles at 2.4 GHz",
      f"  MFMA utilisation, whole chip  = {busy:.0f} / ({v4_us*1e-6*CLK:.0f} x {CUS*SIMDS} SIMDs) = {busy/(v4_us*1e-6*CLK*CUS*SIMDS):.3f}",
      f"  per consumer SIMD ({wgs} workgroups x 4 consumer waves): {n_mfma/(wgs*4):.0f} MFMAs x 32 = {n_mfma/(wgs*4)*32:.0f} busy cycles"
      f" of {v4_us*1e-6*CLK:.0f} = {n_mfma/(wgs*4)*32/(v4_us*1e-6*CLK):.3f}",
      f"  flops {flops/1e9:.2f} G / {v4_us} us = {flops/v4_us/1e6:.0f} TFLOP/s = "
      f"{flops/v4_us/1e6/2500:.3f} of the 2.5 PFLOP/s dense bf16 peak (the kernel is HBM bound: roofline.bound = hbm)",
      ""]
L.append("== gather-bound kernels (tools/neg_pmc.py: kge_score_neg, E=40943 d=512 f32, 512 positives x 1000 negatives)")
tr = sqlite3.connect(f"{out_dir}/neg_trace/r_results.db")
dur = {r[0].split("(")[0].replace("void ", ""): (r[1], r[3]) for r in tr.execute("select * from top_kernels")}
f, w = counters("neg_FETCH_SIZE"), counters("neg_WRITE_SIZE")
names = {"0": "ComplEx", "2": "TransE", "3": "RotatE"}
alg = 512 * 1000 * (512 * 4 + 8 + 4)
L.append("  FETCH_SIZE in KiB, x2 on gfx950 (128-B requests counted as 64 B: MI355X_MICROARCH.md); WRITE_SIZE in KiB")
L.append(f"  algorithmic bytes per launch: 512 x 1000 x (2048 row + 8 index + 4 score) = {alg/1e6:.1f} MB")
for (k, c), (n, v) in sorted(f.items()):
    if "neg_kernel" not in k:
        continue
    fb = v * 1024 * 2
    wb = w[(k, "WRITE_SIZE")][1] * 1024
    cnt, us = dur[k]
    model = names.get(k.split("<")[1].split(",")[0], "?")
    L.append(f"  {model:8s} {k[:44]:44s} avg {us:7.1f} us  fetch {fb/1e6:7.1f} MB  write {wb/1e6:5.2f} MB  "
             f"traffic {(fb+wb)/us/1e6:5.2f} TB/s  = {(fb+wb)/us/1e6/8.0:.2f} of 8 TB/s   algorithmic {alg/us/1e6:5.2f} TB/s")
L += ["  (the 84 MB entity table fits the 256 MB Infinity Cache, so part of this fabric traffic is served",
      "   from MALL rather than HBM; TCC counters sit at the L2 -> fabric boundary and cannot tell the two apart)"]
open(f"profiles/{tag}_rocprofv3_pmc_mfma_gather.txt", "w").write("\n".join(L) + "\n")
print("\n".join(L))
