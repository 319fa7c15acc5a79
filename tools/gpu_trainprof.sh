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
import csv, glob
for f in glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    # one adagrad_multi_kernel launch per step (eager and replayed alike): the step count of the trace
    steps = sum(int(r["Calls"]) for r in rows if "adagrad_multi_kernel" in r.get("Name", ""))
    total_us = sum(float(r.get("TotalDurationNs", 0)) for r in rows) / 1e3
    with open("$OUT/kernel_stats.txt", "w") as o:
        if steps:
            line = f"# {steps} steps in the trace, {total_us / steps:.1f} us of kernels per step, {len(rows)} kernel names"
            print(line); o.write(line + "\n")
        for r in rows[:25]:
            line = f"{r.get('Name', '')[:110]:110s} calls {r.get('Calls'):>6s} avg_ns {float(r.get('AverageNs', 0)):10.0f} pct {r.get('Percentage')}"
            print(line); o.write(line + "\n")
PY
