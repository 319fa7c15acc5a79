#!/bin/bash
# the 1vsAll step after the launch merges: the tests of the touched paths, then the kernel split of the step
#   bash tools/gpu_r5train.sh <tag>
set -u
TAG=${1:-r5train}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ce.py tests/test_gpu_optim.py tests/test_gpu_train_graph.py \
    tests/test_gpu_sharded_two_ranks.py tests/test_gpu_sharded_train.py tests/test_gpu_libkge_plugin.py tests/test_gpu_fuzz_shapes.py \
    -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit: $?" | tee $OUT/env.log
tail -n 25 $OUT/pytest.log
bash tools/gpu_trainprof.sh $TAG 2>&1 | tail -40
