#!/bin/bash
# round-4 profiles on one lease: kernel stats of the bench command, HBM counters of the group launch, matrix-pipe
# counters of the counting kernel       bash tools/gpu_r4prof.sh <tag>
set -u
TAG=${1:-r4prof}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# (1) per-kernel time of the bench command (the short form: main timed regions + legs)
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kstats -o bench -- python $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kstats.err
echo "kernel stats exit: $?" > $OUT/env.log
# (2) HBM bytes of the group launch, separate passes
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C -d $OUT/pmc_$C -o v8 -- python $R/tools/v8_pmc_target.py > /dev/null 2> $OUT/pmc_$C.err
  echo "pmc $C exit: $?" >> $OUT/env.log
done
# (3) the counting kernel on a Wikidata5M shard: matrix-pipe busy cycles, LDS
run() { timeout 120 rocprofv3 --pmc $2 -d $OUT/rank_$1 -o r -- python $R/tools/rank_pmc.py > $OUT/rank_$1.out 2> $OUT/rank_$1.err; echo "rank $1 exit $?" >> $OUT/env.log; }
run a "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
run b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
cd $R
cat $OUT/env.log
python tools/pmc_summary4.py gpurun_out/$TAG 8
python - <<PY
import glob, sqlite3
out = "gpurun_out/$TAG"
with open(out + "/rank_pmc.txt", "w") as f:
    def P(*a):
        s = " ".join(str(x) for x in a); print(s); f.write(s + "\n")
    P("rocprofv3 --pmc (separate passes) over tools/rank_pmc.py: 30 launches of pairs_bf16_v8_rank_kernel<ComplEx, 128, 0> "
      "(d = 256: two accumulators per chain), n = 512 x 2 directions, E = 574,311, no filter sets")
    vals = {}
    for sub in ("a", "b"):
        for db in glob.glob(f"{out}/rank_{sub}/**/*_results.db", recursive=True):
            con = sqlite3.connect(db)
            for r in con.execute("select counter_name, count(*), avg(value) from counters_collection "
                                 "where kernel_name like '%pairs_bf16_v8_rank_kernel%' group by counter_name"):
                P(sub, r[0], "dispatches=%d mean/dispatch=%.1f" % (r[1], r[2])); vals[r[0]] = r[2]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_BUSY_CU_CYCLES" in vals:
        P("matrix-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) = %.3f"
          % (vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * vals["SQ_BUSY_CU_CYCLES"])))
    # kernel stats of the bench command
    for db in glob.glob(f"{out}/kstats/**/*_results.db", recursive=True):
        con = sqlite3.connect(db)
        rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
        with open(out + "/kernel_stats.txt", "w") as g:
            g.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 40 --warmup 8 --no-cpu-baseline (durations in us)\n")
            g.write(f"{'kernel':122s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'pct':>6s}\n")
            for r in rows[:40]:
                g.write(f"{r[0][:120]:122s} {r[1]:>6} {float(r[2]):11.1f} {float(r[3]):9.2f} {float(r[4]):6.2f}\n")
        print(open(out + "/kernel_stats.txt").read()[:3000])
PY
