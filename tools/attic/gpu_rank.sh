#!/bin/bash
# counting-epilogue iteration: the fused rank tests + the evaluation tests   bash tools/gpu_rank.sh <tag>
set -u
TAG=${1:-rank}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_score_rank.py tests/test_gpu_model_eval.py tests/test_gpu_parity.py tests/test_gpu_optim.py -m gpu -q -x --timeout=600 > $OUT/pytest_rank.log 2>&1
echo "pytest exit: $?" > $OUT/env.log
tail -n 25 $OUT/pytest_rank.log
cat $OUT/env.log
