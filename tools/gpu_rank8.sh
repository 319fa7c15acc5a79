#!/bin/bash
# iteration on the counting kernel   bash tools/gpu_rank8.sh <tag>
set -u
TAG=${1:-rank8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo > $OUT/env.log
timeout 1200 python -m pytest tests/test_gpu_score_rank.py -m gpu -q -x --timeout=900 > $OUT/pytest_rank.log 2>&1
echo "pytest rank exit: $?" >> $OUT/env.log
tail -n 30 $OUT/pytest_rank.log
timeout 600 python tools/rank8_probe.py > $OUT/rank8_probe.txt 2>&1
echo "probe exit: $?" >> $OUT/env.log
cat $OUT/env.log
cat $OUT/rank8_probe.txt
