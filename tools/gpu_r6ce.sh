#!/bin/bash
# round 6: the fused-loss passes on the persistent kernel -- the CE tests, then the kernel split of the step
#   bash tools/gpu_r6ce.sh <tag>
set -u
TAG=${1:-r6ce}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ce.py tests/test_gpu_train_graph.py tests/test_gpu_fuzz_shapes.py -m gpu -q -x --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit: $?" | tee $OUT/env.log
tail -n 15 $OUT/pytest.log
bash tools/gpu_trainprof.sh $TAG 2>&1 | tail -40
