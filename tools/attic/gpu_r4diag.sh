#!/bin/bash
# round 4, first lease: what bounds the direct-store launch in steady state   bash tools/gpu_r4diag.sh <tag>
set -u
TAG=${1:-r4diag}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo > $OUT/env.log
timeout 300 python -m pytest tests/test_gpu_queries.py -m gpu -q -x --timeout=300 > $OUT/pytest_queries.log 2>&1
echo "pytest queries exit: $?" >> $OUT/env.log
tail -n 5 $OUT/pytest_queries.log
timeout 300 ./tools/ubench/store_rate > $OUT/store_rate.txt 2>&1
echo "store_rate exit: $?" >> $OUT/env.log
timeout 600 python tools/r4_diag.py > $OUT/r4_diag.txt 2>&1
echo "diag exit: $?" >> $OUT/env.log
cat $OUT/env.log
cat $OUT/store_rate.txt
cat $OUT/r4_diag.txt
