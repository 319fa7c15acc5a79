#!/bin/bash
# iteration on the persistent two-consumers-per-SIMD kernel   bash tools/gpu_v8.sh <tag> [pytest -k expr]
set -u
TAG=${1:-v8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo > $OUT/env.log
timeout 900 python -m pytest tests/test_gpu_queries.py -m gpu -q -x --timeout=600 ${2:+-k "$2"} > $OUT/pytest_queries.log 2>&1
echo "pytest queries exit: $?" >> $OUT/env.log
tail -n 30 $OUT/pytest_queries.log
timeout 600 python tools/v8_probe.py > $OUT/v8_probe.txt 2>&1
echo "probe exit: $?" >> $OUT/env.log
cat $OUT/env.log
cat $OUT/v8_probe.txt
