#!/bin/bash
# round 6, second session: the hip reciprocal wrapper (plugin test d2) + the plugin / evaluation tests around it
#   bash tools/gpu_r6recip.sh <tag>
set -u
TAG=${1:-r6recip}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export KGE_PLUGIN_LOG=$OUT/plugin.jsonl
timeout 1500 python -m pytest tests/test_gpu_libkge_plugin.py -m gpu -q -x --timeout=900 -p no:cacheprovider -k "${KEXPR:-d2 or test_d_ or c4 or test_g_}" > $OUT/pytest.log 2>&1
echo "pytest exit: $?" | tee $OUT/env.log
tail -n 30 $OUT/pytest.log | grep -v "Warning\|warn\|^$\|labels = \|jit" | tail -22
grep "d2:" $OUT/plugin.jsonl | cut -c1-400
