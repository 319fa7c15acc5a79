#!/usr/bin/env python3
"""tools/gpu_r3neg.sh's passes -> profiles/<tag>_rocprofv3_neg.txt + profiles/pmc_neg_latest.json (read by bench.py's
roofline_neg leg for `traffic`).   python tools/neg_pmc_summary.py gpurun_out/<tag> <tag>"""
import json
import os
import sqlite3
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
n, K, d = 512, 1000, 512
names = {"2": "transe", "3": "rotate", "0": "complex", "1": "distmult"}


def counters(sub):
    c = sqlite3.connect(f"{out_dir}/{sub}/r_results.db")
    res = {}
    for k, cn, cnt, v in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by 1,2"):
        if "neg_kernel" in k:
            res[names.get(k.split("<")[1].split(",")[0].strip(), "?")] = (cnt, v)
    return res


def durations(sub):
    c = sqlite3.connect(f"{out_dir}/{sub}/r_results.db")
    res = {}
    for r in c.execute("select * from top_kernels"):
        if "neg_kernel" in r[0]:
            res[names.get(r[0].split("<")[1].split(",")[0].strip(), "?")] = (r[1], r[3])
    return res


L = [f"{tag}: kge_score_neg (neg_kernel), 512 positives x 1000 negatives, d = 512, float32 -- rocprofv3 passes of tools/neg_pmc.py",
     "kernel-trace and each --pmc counter in a run of its own; FETCH_SIZE in KiB x2 on gfx950 (128-B requests tallied at 64 B:",
     "MI355X_MICROARCH.md), WRITE_SIZE in KiB; algorithmic bytes per launch = n K (d 4 + 4 + 8) + fixed rows", ""]
js = {}
for case, E in (("wn18rr", 40943), ("big", 2000000)):
    key = "wn18rr" if case == "wn18rr" else "beyond_infinity_cache"
    try:
        dur, f, w = durations(f"neg_{case}_trace"), counters(f"neg_{case}_FETCH_SIZE"), counters(f"neg_{case}_WRITE_SIZE")
    except Exception as e:  # a pass is missing
        L.append(f"{case}: {e}")
        continue
    L.append(f"== E = {E:,} ({E * d * 4 / 1e6:.0f} MB table{': inside' if case == 'wn18rr' else ': beyond'} the 256 MB Infinity Cache)")
    for model in ("rotate", "transe"):
        if model not in dur:
            continue
        dr = d // 2 if model == "rotate" else d
        alg = n * K * (d * 4 + 4 + 8) + n * (d + dr + d) * 4 + 3 * n * 8
        calls, us = dur[model]
        fb, wb = f[model][1] * 1024 * 2, w[model][1] * 1024
        js[f"{key}_{model}"] = fb + wb
        L.append(f"  {model:7s} avg {us:7.1f} us over {calls} launches | fetch {fb / 1e6:8.1f} MB write {wb / 1e6:5.2f} MB = "
                 f"{(fb + wb) / alg:.3f} x the {alg / 1e6:.1f} MB algorithmic | traffic {(fb + wb) / us / 1e6:.2f} TB/s = "
                 f"{(fb + wb) / us / 1e6 / 8:.3f} of 8 TB/s, algorithmic {alg / us / 1e6:.2f} TB/s = {alg / us / 1e6 / 8:.3f}")
    L.append("")
os.makedirs("profiles", exist_ok=True)
open(f"profiles/{tag}_rocprofv3_neg.txt", "w").write("\n".join(L) + "\n")
json.dump(js, open("profiles/pmc_neg_latest.json", "w"), indent=1)
print("\n".join(L))
