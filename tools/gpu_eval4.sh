#!/bin/bash
# kge_eval_batch: tests, then a kernel trace of tools/eval_probe.py (launches per batch, share of the counting kernel)
set -u
TAG=${1:-ev4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_score_rank.py tests/test_gpu_model_eval.py -m gpu -q -x --timeout=600 > $OUT/pytest.log 2>&1
echo "pytest exit: $?" > $OUT/env.log
tail -n 6 $OUT/pytest.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o ev -- python $R/tools/eval_probe.py > $R/$OUT/eval_probe.txt 2>&1
KGE_EVAL_LAUNCH_BY_LAUNCH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_old -o ev -- python $R/tools/eval_probe.py > $R/$OUT/eval_probe_old.txt 2>&1
cd $R
cat $OUT/env.log
grep "index build" $OUT/eval_probe.txt $OUT/eval_probe_old.txt
python tools/db_summary.py $OUT 2>&1 | head -40
