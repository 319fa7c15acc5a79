#!/bin/bash
# the negative-sampling backward sorted by entity: tests, the bench leg, the step's kernel split (round 5)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5ns}
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_model_eval.py tests/test_gpu_fuzz_shapes.py tests/test_gpu_libkge_plugin.py -m gpu -q -p no:cacheprovider -k "score_neg or negative or fuzz or test_b" 2>&1 | tail -4 | grep -v Warn
timeout 300 python -c "
import json, torch, bench
print(json.dumps(bench.ns_step_leg(torch.device('cuda', 0), 512, 20)))" 2>&1 | grep -v amdgpu | tee $OUT/ns_leg.txt
cd /tmp
MODEL=rotate E=40943 R=11 D=512 K=1000 STEPS=20 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/rotate -o ns -- python $GRAFT_REPO_ROOT/tools/ns_step_kernels.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import glob, sqlite3
for db in glob.glob("$OUT/rotate/**/*_results.db", recursive=True):
    con = sqlite3.connect(db)
    for r in list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))[:14]:
        print(f"  {r[0][:110]:110s} {r[1]:>5} {float(r[2]):10.1f} {float(r[3]):9.2f} {float(r[4]):6.2f}")
PY
