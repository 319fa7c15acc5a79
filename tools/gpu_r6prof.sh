#!/bin/bash
# round-6 profiles on one lease: the driver-like bench line, kernel stats of the bench command, HBM counters of the group
# launch (separate passes), matrix-pipe counters of the counting kernel, the training step's kernel split, the one-rank
# RCCL leg                                   bash tools/gpu_r6prof.sh <tag>
set -u
TAG=${1:-r6prof}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" > $OUT/env.log
timeout 900 env KGE_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err
echo "bench dist1 exit: $?" >> $OUT/env.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kstats -o bench -- python $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kstats.err
echo "kernel stats exit: $?" >> $OUT/env.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C -d $OUT/pmc_$C -o v8 -- python $R/tools/v8_pmc_target.py > /dev/null 2> $OUT/pmc_$C.err
  echo "pmc $C exit: $?" >> $OUT/env.log
done
run() { timeout 120 rocprofv3 --pmc $2 -d $OUT/rank_$1 -o r -- python $R/tools/rank_pmc.py > $OUT/rank_$1.out 2> $OUT/rank_$1.err; echo "rank $1 exit $?" >> $OUT/env.log; }
run a "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
cd $R
PMC_OUT=$OUT PMC_TXT=pmc_hbm.txt PMC_SOURCE=profiles/r6_rocprofv3_pmc_hbm.txt python tools/pmc_summary4.py gpurun_out/$TAG 8 > /dev/null
cat $OUT/pmc_hbm.txt | tail -4
python - <<PY
import glob, sqlite3
out = "gpurun_out/$TAG"
with open(out + "/rank_pmc.txt", "w") as f:
    def P(*a):
        s = " ".join(str(x) for x in a); print(s); f.write(s + "\n")
    P("rocprofv3 --pmc (one pass) over tools/rank_pmc.py: 30 launches of pairs_bf16_v8_rank_kernel<ComplEx, 128, 0> "
      "(d = 256: two accumulators per chain), n = 512 x 2 directions, E = 574,311, no filter sets")
    vals = {}
    for db in glob.glob(f"{out}/rank_a/**/*_results.db", recursive=True):
        con = sqlite3.connect(db)
        for r in con.execute("select counter_name, count(*), avg(value) from counters_collection "
                             "where kernel_name like '%pairs_bf16_v8_rank_kernel%' group by counter_name"):
            P(r[0], "dispatches=%d mean/dispatch=%.1f" % (r[1], r[2])); vals[r[0]] = r[2]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_BUSY_CU_CYCLES" in vals:
        P("matrix-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) = %.3f"
          % (vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * vals["SQ_BUSY_CU_CYCLES"])))
for db in glob.glob(f"{out}/kstats/**/*_results.db", recursive=True):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out + "/kernel_stats.txt", "w") as g:
        g.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 40 --warmup 8 --no-cpu-baseline (durations in us)\n")
        g.write(f"{'kernel':122s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'pct':>6s}\n")
        for r in rows[:45]:
            g.write(f"{r[0][:120]:122s} {r[1]:>6} {float(r[2]):11.1f} {float(r[3]):9.2f} {float(r[4]):6.2f}\n")
    print(open(out + "/kernel_stats.txt").read()[:2500])
PY
# the counting kernel in its three forms (split, single-pass, band-and-rescore) under the kernel trace
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/band -o band -- python $R/tools/rank_band_bench.py 20 > $OUT/rank_band_bench.txt 2> $OUT/band.err
echo "band stats exit: $?" >> $OUT/env.log
cd $R
python - <<PY
import glob, sqlite3
out = "gpurun_out/$TAG"
for db in glob.glob(f"{out}/band/**/*_results.db", recursive=True):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out + "/rank_band_kernel_stats.txt", "w") as g:
        g.write("# rocprofv3 --kernel-trace --stats -- python tools/rank_band_bench.py 20 (durations in us): the counting kernel as\n"
                "# pairs_bf16_v8_rank_kernel<.., 128|256, SPLIT, 0, BAND> -- <.,.,1,0,0> split, <.,.,0,0,0> single-pass, <.,.,0,0,1> the band launch --,\n"
                "# pairs_bf16_rescore_kernel = the band's second launch; FB15k-237 shape (d = 512) and Wikidata5M shard (d = 256) mixed\n")
        g.write(f"{'kernel':122s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'pct':>6s}\n")
        for r in rows[:24]:
            g.write(f"{r[0][:120]:122s} {r[1]:>6} {float(r[2]):11.1f} {float(r[3]):9.2f} {float(r[4]):6.2f}\n")
    print(open(out + "/rank_band_kernel_stats.txt").read()[:1800])
PY
bash tools/gpu_trainprof.sh $TAG/train > /dev/null 2>&1
cp gpurun_out/$TAG/train/kernel_stats.txt $OUT/train_kernel_stats.txt 2>/dev/null
cat gpurun_out/$TAG/train/wall.txt 2>/dev/null | tail -5
cat $OUT/env.log
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print(json.dumps({k: d[k] for k in ("value", "value_settled", "ms_per_step", "value_kind")}))
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "traffic", d["roofline"]["traffic"])
print("training", d["training_tolerance"]["roofline"]["frac"])
print("train", json.dumps(d.get("roofline_train"))[:600])
print("cpu", d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"])
PY
# the raw traces stay on the box (gpurun merges at most 64 MiB back): the summaries above are what travels
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -type f -size +2M -delete 2>/dev/null
du -sh $OUT
