#!/bin/bash
# round 6: penalty terms folded into HipAdagrad -- the optimizer tests, the plugin's captured step with penalties, the
# sharded jobs on the GPU     bash tools/gpu_plugin.sh --timeout 1500 -- 'bash tools/gpu_r6pen.sh <tag>'
set -u
TAG=${1:-r6pen}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
KGE_PLUGIN_LOG=$OUT/plugin.jsonl timeout 1200 python -m pytest tests/test_gpu_optim.py tests/test_gpu_libkge_plugin.py -m gpu -q -x --timeout=900 -p no:cacheprovider -k "${KEXPR:-optim or test_a_ or test_l_ or test_j_ or test_f_}" > $OUT/pytest.log 2>&1
echo "pytest exit: $?" | tee $OUT/env.log
tail -n 25 $OUT/pytest.log
