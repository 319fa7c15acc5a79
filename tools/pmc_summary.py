#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes) of bench.py into
profiles/<tag>_rocprofv3_pmc_hbm.txt and profiles/pmc_latest.json (read by bench.py)."""
import json
import sqlite3
import statistics
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = sqlite3.connect(f"{out_dir}/pmc_{c}/bench_results.db")
    per_kernel = {}
    for name, v in d.execute("select kernel_name, value from counters_collection where counter_name=?", (c,)):
        per_kernel.setdefault(name.split("(")[0], []).append(v)
    res[c] = {k: (len(v), statistics.mean(v)) for k, v in per_kernel.items()}
lines = [f"rocprofv3 --pmc <C> -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-sided (separate passes), {tag}",
         "values in KiB per dispatch (mean); gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM)"]
tot = 0.0
for c in res:
    for k, (n, m) in res[c].items():
        lines.append(f"{c:11s} {k[:60]:60s} dispatches={n:4d} mean={m:10.1f} KiB")
main = [k for k in res["FETCH_SIZE"] if "pairs_bf16_v7_kernel" in k] or \
       [k for k in res["FETCH_SIZE"] if "pairs_bf16_v6_kernel" in k] or \
       [k for k in res["FETCH_SIZE"] if "pairs_bf16_v4_kernel" in k] or \
       [k for k in res["FETCH_SIZE"] if "pairs_bf16_v3_kernel" in k] or \
       [k for k in res["FETCH_SIZE"] if "pairs_bf16_v2_kernel" in k]
if main:
    k = main[0]
    fetch = res["FETCH_SIZE"][k][1] * 1024 * 2
    write = res["WRITE_SIZE"][k][1] * 1024
    bq = [x for x in res["FETCH_SIZE"] if "build_queries" in x]
    extra = 0.0
    if bq:
        extra = res["FETCH_SIZE"][bq[0]][1] * 1024 * 2 + res["WRITE_SIZE"][bq[0]][1] * 1024
    tot = fetch + write + extra
    try:  # the bench line of the same round says what a launch is and its algorithmic bytes
        bj = json.loads([ln for ln in open(f"{out_dir}/bench.json") if ln.startswith("{")][-1])
        alg = bj["roofline"]["algorithmic_bytes_per_launch"]
        launch = "score_sp_po" if "two-sided" in bj["roofline"]["kernel"] else "score_sp"
    except Exception:
        alg, launch = 45726720, "score_sp"
    lines.append(f"per launch ({launch}): fetch(corrected) {fetch/1e6:.2f} MB + write {write/1e6:.2f} MB + builder {extra/1e6:.2f} MB = {tot/1e6:.2f} MB; algorithmic {alg/1e6:.2f} MB")
    json.dump({"hbm_bytes_per_launch": tot, "fetch_bytes_corrected": fetch, "write_bytes": write,
               "builder_bytes": extra, "launch": launch, "source": f"profiles/{tag}_rocprofv3_pmc_hbm.txt"},
              open("profiles/pmc_latest.json", "w"))
open(f"profiles/{tag}_rocprofv3_pmc_hbm.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
