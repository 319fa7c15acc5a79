#!/bin/bash
# round 6, second session: target ranges (kge_index.start), the reference's entity_ranking job over a hip model with
# split queries under no_grad, the options that replaced the environment toggles
#   bash tools/gpu_r6range.sh <tag>
set -u
TAG=${1:-r6range}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_queries.py tests/test_gpu_libkge_plugin.py tests/test_gpu_score_rank.py tests/test_gpu_model_eval.py tests/test_gpu_parity.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit: $?" | tee $OUT/env.log
tail -n 25 $OUT/pytest.log
