#!/bin/bash
# round 6, second session: split queries on the d = 256 persistent store kernel
#   bash tools/gpu_r6fork.sh <tag>
set -u
TAG=${1:-r6fork}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_queries.py tests/test_gpu_ce.py tests/test_gpu_train_graph.py tests/test_gpu_bwd_gemm16.py -m gpu -q -x --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit: $?" | tee $OUT/env.log
tail -n 8 $OUT/pytest.log
for sws in "" "BWD_FORK=0" "" "BWD_FORK=0"; do
  echo "== KGE_SWITCHES=$sws" | tee -a $OUT/step.log
  KGE_SWITCHES=$sws ONLY=bf16_scoring STEPS=300 timeout 300 python tools/train_step_prof.py 2>&1 | tee -a $OUT/step.log
done
