#!/usr/bin/env python3
"""Copy what a tools/gpu_round.sh run left under gpurun_out/<tag> (+ <tag>pmc) into profiles/<tag>_*:
bench lines, probes, the rocprofv3 kernel-trace summary (top_kernels view of the result database),
the HBM counter summary (pmc_summary.py) and the MFMA / gather counter summary (pmc_extra.py).
Usage: python tools/collect_profiles.py <tag>"""
import glob
import os
import shutil
import sqlite3
import subprocess
import sys

tag = sys.argv[1]
src, dst = f"gpurun_out/{tag}", "profiles"
for a, b in (("bench.json", "bench.json"), ("bench_dist1.json", "bench_dist1rank.json"), ("probe.jsonl", "probe.jsonl"),
             ("v2_phases.txt", "phase_timestamps.txt"), ("bwd_probe.txt", "bwd_probe.txt"),
             ("ns_probe.txt", "ns_probe.txt"), ("train_probe.txt", "train_probe.txt"),
             ("eval_probe.txt", "eval_probe.txt"), ("subset_probe.txt", "subset_probe.txt"),
             ("ce_probe.txt", "ce_probe.txt"), ("ce_phases.txt", "ce_phases.txt"), ("ce_host.txt", "ce_host.txt")):
    if os.path.exists(f"{src}/{a}"):
        if a.endswith(".json"):  # RCCL prints a version banner to stdout before the bench line
            rows = [ln for ln in open(f"{src}/{a}") if ln.startswith("{")]
            open(f"{dst}/{tag}_{b}", "w").write("".join(rows))
        else:
            shutil.copy(f"{src}/{a}", f"{dst}/{tag}_{b}")
dbs = glob.glob(f"{src}/prof/**/*_results.db", recursive=True)
v4_us = None
if dbs:
    c = sqlite3.connect(dbs[0])
    lines = [f"rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-one-sided  (MI355X, round 1 run {tag})",
             "name | total_calls | total_duration | average | percentage"]
    for r in c.execute("select * from top_kernels"):
        lines.append(" | ".join(str(x) for x in r))
        if "pairs_bf16_v4_kernel" in r[0]:
            v4_us = r[3]
    open(f"{dst}/{tag}_rocprofv3_kernel_stats.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:4]))
subprocess.check_call([sys.executable, "tools/pmc_summary.py", src, tag])
if os.path.isdir(f"gpurun_out/{tag}pmc"):
    subprocess.check_call([sys.executable, "tools/pmc_extra.py", f"gpurun_out/{tag}pmc", tag] +
                          ([f"{v4_us:.1f}"] if v4_us else []))
