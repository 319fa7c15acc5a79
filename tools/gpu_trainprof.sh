#!/bin/bash
# kernel split of the training step   bash tools/gpu_trainprof.sh <tag>
set -u
TAG=${1:-trainprof}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/train_step_prof.py 2>&1 | grep -v amdgpu.ids | tee $OUT/wall.txt
cd /tmp
ONLY=bf16_scoring STEPS=100 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o train -- python $GRAFT_REPO_ROOT/tools/train_step_prof.py > $OUT/prof.log 2>&1
echo "rocprof exit: $?"
cd $GRAFT_REPO_ROOT
python - <<PY
import glob, sqlite3
for db in glob.glob("$OUT/prof/**/*_results.db", recursive=True):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    # one adagrad_multi_kernel launch per step (eager and replayed alike): the step count of the trace
    steps = sum(int(r[1]) for r in rows if "adagrad_multi" in r[0])
    total_us = sum(float(r[2]) for r in rows)
    with open("$OUT/kernel_stats.txt", "w") as o:
        def P(line):
            print(line); o.write(line + "\n")
        P("# rocprofv3 --kernel-trace --stats -- ONLY=bf16_scoring STEPS=100 python tools/train_step_prof.py (durations in us):")
        P("# the fused 1vsAll step at the FB15k-237 shape (n = 512, ComplEx d = 512, bf16 scoring copies, HipAdagrad), eager steps and graph replays mixed")
        if steps:
            P(f"# {steps} steps in the trace, {total_us / steps:.1f} us of kernels per step, {len(rows)} kernel names")
        for r in rows[:25]:
            P(f"{r[0][:110]:110s} calls {r[1]:>6} avg_us {float(r[3]):8.2f} pct {float(r[4]):6.2f}")
PY
