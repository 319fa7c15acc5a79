#!/usr/bin/env python3
"""Per launch SHAPE statistics of one kernel instantiation from a rocprofv3 --kernel-trace database (rocpd
`*_results.db`): durations in time order, clusters by duration, end -> start gaps between consecutive launches.  The
`--stats` table averages over every launch of an instantiation -- for pairs_bf16_v8_kernel that mixes the bench step's
two-sided groups with the one-sided groups of another leg (the header of profiles/r4_rocprofv3_kernel_stats.txt).

    python tools/trace_shapes.py gpurun_out/r4prof2/kstats/bench_results.db pairs_bf16_v8_kernelILi0ELi0ELi2EE 95
                                 (database)                                  (substring of the mangled name)  (cut, us)"""
import sqlite3, statistics as st, sys

db, name, cut = sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else None
c = sqlite3.connect(db)
rows = list(c.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start"))
seq = [(s, e) for s, e, n in rows if name in n]
d = [(e - s) / 1e3 for s, e in seq]
print(f"{name}: {len(d)} launches, mean {st.mean(d):.1f} us")
if cut is not None:
    for tag, part in (("at or above", [x for x in d if x >= cut]), ("below", [x for x in d if x < cut])):
        if part:
            print(f"  {tag} {cut:g} us: n = {len(part)}  mean {st.mean(part):.1f}  median {st.median(part):.1f}  "
                  f"min {min(part):.1f}  max {max(part):.1f}")
    print("  in time order (at or above the cut):", " ".join(str(int(x)) for x in d if x >= cut))
gaps = [(s1 - e0) / 1e3 for (s0, e0), (s1, e1) in zip(seq, seq[1:]) if (s1 - e0) / 1e3 < 50]
if gaps:
    print(f"  end -> start between consecutive launches (those under 50 us): n = {len(gaps)}  median {st.median(gaps):.2f}  "
          f"max {max(gaps):.1f} us")
