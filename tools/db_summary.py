#!/usr/bin/env python3
"""Print per-kernel averages from every rocprofv3 result database under a gpurun_out/<tag> directory:
kernel-trace runs (top_kernels view) and --pmc runs (mean counter value per dispatch; FETCH_SIZE /
WRITE_SIZE are KiB, FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md)."""
import glob
import os
import sqlite3
import statistics
import sys

root = sys.argv[1]
for db in sorted(glob.glob(f"{root}/**/*_results.db", recursive=True)):
    rel = os.path.relpath(db, root)
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, value from counters_collection"))
    except sqlite3.Error:
        rows = []
    if rows:
        agg = {}
        for k, cn, v in rows:
            agg.setdefault((cn, k.split("(")[0][:70]), []).append(v)
        for (cn, k), v in sorted(agg.items()):
            if any(x in k for x in ("pairs", "rank", "neg", "ce_", "gemm16", "adagrad")):
                m = statistics.mean(v)
                extra = f" -> {m * 1024 * (2 if cn == 'FETCH_SIZE' else 1) / 1e6:.2f} MB" if cn in ("FETCH_SIZE", "WRITE_SIZE") else ""
                print(f"{rel}: {cn} {k} n={len(v)} mean={m:.1f}{extra}")
        continue
    try:
        for r in c.execute("select * from top_kernels"):
            if r[4] >= 1.0:
                print(f"{rel}: {r[0][:80]} calls={r[1]} avg_us={r[3]:.2f} pct={r[4]:.1f}")
    except sqlite3.Error as e:
        print(rel, "no top_kernels:", e)
