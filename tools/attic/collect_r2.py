#!/usr/bin/env python3
"""Copy what tools/gpu_r2.sh left under gpurun_out/<tag> into profiles/<tag>_*: the LibKGE-plugin and
FB15k-237-shape rank logs of the GPU tests, the bench line, the rocprofv3 kernel-trace summaries (two-sided
bench launches, one-sided launches, one-sided with padded pitch), the FETCH_SIZE / WRITE_SIZE counter
summary (+ profiles/pmc_latest.json, read by bench.py) and the phase stamps.
Usage: python tools/collect_r2.py <tag>"""
import glob
import json
import os
import shutil
import sqlite3
import statistics
import sys

tag = sys.argv[1]
src, dst = f"gpurun_out/{tag}", "profiles"
for a, b in (("plugin_gpu.jsonl", "libkge_plugin_gpu.jsonl"), ("bshape_ranks.jsonl", "bshape_ranks.jsonl"),
             ("v2_phases.txt", "phase_timestamps.txt"), ("env.log", "env.log"),
             ("gemm16_probe.txt", "gemm16_probe.txt"), ("gemm16_phases.txt", "gemm16_phase_stamps.txt"),
             ("bench_dist1_wikidata5m.json", "bench_dist1rank_wikidata5m.json"),
             ("bench_dist1_fb15k.json", "bench_dist1rank_fb15k.json")):
    if os.path.exists(f"{src}/{a}"):
        shutil.copy(f"{src}/{a}", f"{dst}/{tag}_{b}")
# the rank-parity log holds the FB15k-237-shape lines and (tagged "shape": "wn18rr") the WN18RR-shape ones
if os.path.exists(f"{src}/bshape_ranks.jsonl"):
    lines = open(f"{src}/bshape_ranks.jsonl").read().splitlines()
    open(f"{dst}/{tag}_bshape_ranks.jsonl", "w").write("".join(l + "\n" for l in lines if '"wn18rr"' not in l))
    open(f"{dst}/{tag}_wshape_ranks.jsonl", "w").write("".join(l + "\n" for l in lines if '"wn18rr"' in l))
if os.path.exists(f"{src}/bench.json"):
    rows = [ln for ln in open(f"{src}/bench.json") if ln.startswith("{")]
    open(f"{dst}/{tag}_bench.json", "w").write("".join(rows))
if os.path.exists(f"{src}/pytest_all.log"):
    tail = [ln for ln in open(f"{src}/pytest_all.log") if " passed" in ln or " failed" in ln or ln.startswith("FAILED")]
    open(f"{dst}/{tag}_pytest_gpu_summary.txt", "w").write("pytest tests -m gpu (MI355X, reference package shipped for the plugin tests)\n" + "".join(tail))
lines = [f"rocprofv3 --kernel-trace --stats, MI355X, run {tag}: name | calls | total (us) | average (us) | % of run"]
for sub, what in (("prof", "python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-one-sided  (two-sided score_sp_po launches)"),
                  ("prof1", "python tools/one_sided.py  (one-sided score_sp launches, C2 shape)"),
                  ("prof1p", "python tools/one_sided.py --pad  (one-sided, row pitch padded to 32 floats)"),
                  ("prof_step", "python tools/step_kernels.py  (30 fused 1vsAll training steps: loss_sp_po + backward + one-pass Adagrad)")):
    dbs = glob.glob(f"{src}/{sub}/**/*_results.db", recursive=True)
    if not dbs:
        continue
    lines.append(f"-- {what}")
    for r in sqlite3.connect(dbs[0]).execute("select * from top_kernels"):
        lines.append(" | ".join(str(x) for x in r))
open(f"{dst}/{tag}_rocprofv3_kernel_stats.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:4]))


def counter(sub, name):
    dbs = glob.glob(f"{src}/{sub}/**/*_results.db", recursive=True)
    if not dbs:
        return None
    v = [x[0] for x in sqlite3.connect(dbs[0]).execute(
        "select value from counters_collection where counter_name=? and kernel_name like '%pairs_bf16_v4_kernel%'", (name,))]
    return (len(v), statistics.mean(v)) if v else None


out = [f"rocprofv3 --pmc <C> (separate passes), MI355X, run {tag}; values KiB per dispatch (mean);",
       "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM)"]
summ = {}
for key, what, alg in (("pmc", "two-sided launch (bench.py --no-one-sided)", 76563456),
                       ("pmc1", "one-sided launch (tools/one_sided.py)", 45726720),
                       ("pmc1p", "one-sided launch, padded pitch (tools/one_sided.py --pad)", 45726720)):
    f, w = counter(f"{key}_FETCH_SIZE", "FETCH_SIZE"), counter(f"{key}_WRITE_SIZE", "WRITE_SIZE")
    if not f or not w:
        continue
    fb, wb = f[1] * 1024 * 2, w[1] * 1024
    out.append(f"{what}: FETCH_SIZE {f[1]:.1f} KiB x2 = {fb / 1e6:.2f} MB ({f[0]} dispatches) + WRITE_SIZE {w[1]:.1f} KiB = "
               f"{wb / 1e6:.2f} MB -> {(fb + wb) / 1e6:.2f} MB per launch; algorithmic {alg / 1e6:.2f} MB ({(fb + wb) / alg:.3f}x)")
    summ[key] = (fb, wb)
open(f"{dst}/{tag}_rocprofv3_pmc_hbm.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
if "pmc" in summ:
    fb, wb = summ["pmc"]
    json.dump({"hbm_bytes_per_launch": fb + wb, "fetch_bytes_corrected": fb, "write_bytes": wb, "builder_bytes": 0.0,
               "launch": "score_sp_po", "source": f"profiles/{tag}_rocprofv3_pmc_hbm.txt"},
              open("profiles/pmc_latest.json", "w"))
