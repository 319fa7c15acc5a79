#!/usr/bin/env python3
"""Round 4: summarise the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/v8_pmc_target.py into
profiles/r4_rocprofv3_pmc_hbm.txt and profiles/pmc_latest.json (read by bench.py: traffic per GROUP launch, per query
mode).  gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes -> x 2 (MI355X_MICROARCH.md, HBM section); both
counters are in KiB."""
import json
import sqlite3
import statistics
import sys

out_dir, group = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = sqlite3.connect(f"{out_dir}/pmc_{c}/v8_results.db")
    per = {}
    for name, v in d.execute("select kernel_name, value from counters_collection where counter_name=?", (c,)):
        per.setdefault(name, []).append(v)
    res[c] = per
lines = [f"rocprofv3 --pmc <C> -- python tools/v8_pmc_target.py (separate passes): 12 group launches of {group} two-sided batches "
         "(n = 512, E = 14,541, d = 512) per query mode; KiB per dispatch, median over the dispatches behind the first two",
         "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x 2"]
alg = group * (14541 * 512 * 2 + 2 * (512 * 1024 * 2 + 512 * 14541 * 4 + 2 * 512 * 8))
summary = {"launch": f"score_sp_po_group{group}", "group": group, "algorithmic_bytes_per_launch": alg,
           "source": "profiles/r4_rocprofv3_pmc_hbm.txt"}
for c, per in res.items():
    for k, v in per.items():
        lines.append(f"{c:11s} {k[:100]:100s} dispatches={len(v):3d} median={statistics.median(v):12.1f} KiB")


def pick(c, split):
    tag = "pairs_bf16_v8_kernel<0, %d," % (1 if split else 0)
    ks = [k for k in res[c] if tag in k.replace("(int)", "")]
    if not ks:
        return None
    v = res[c][ks[0]]
    return statistics.median(v[2:] if len(v) > 4 else v) * 1024.0


for mode, split in (("parity", True), ("training", False)):
    f, w = pick("FETCH_SIZE", split), pick("WRITE_SIZE", split)
    if f is None or w is None:
        lines.append(f"{mode}: kernel not found in the passes")
        continue
    f *= 2.0
    b = []
    for c, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):  # the query-build part of the same launch sequence
        for k, v in res[c].items():
            if "build_queries" in k or "query_build" in k:
                b.append(statistics.median(v) * 1024.0 * mul)
    summary[mode] = {"hbm_bytes_per_launch": f + w, "fetch_bytes_corrected": f, "write_bytes": w}
    lines.append(f"{mode}: per launch fetch(corrected) {f / 1e6:.1f} MB + write {w / 1e6:.1f} MB = {(f + w) / 1e6:.1f} MB; "
                 f"algorithmic {alg / 1e6:.1f} MB (score matrix {group * 2 * 512 * 14541 * 4 / 1e6:.1f} MB of it)")
import os
OUTP = os.environ.get("PMC_OUT", "profiles")
summary["source"] = os.environ.get("PMC_SOURCE", summary["source"])
json.dump(summary, open(OUTP + "/pmc_latest.json", "w"))
open(OUTP + "/" + os.environ.get("PMC_TXT", "r4_rocprofv3_pmc_hbm.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
