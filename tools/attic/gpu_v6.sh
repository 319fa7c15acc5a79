#!/bin/bash
# iteration on the unit-pipelined kernel: parity tests of the prepared path, then v6 vs v4   bash tools/gpu_v6.sh <tag> [steps] [rounds]
set -u
TAG=${1:-v6}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo > $OUT/env.log
timeout 600 python -m pytest tests/test_gpu_queries.py -m gpu -q -x --timeout=300 > $OUT/pytest_queries.log 2>&1
echo "pytest queries exit: $?" >> $OUT/env.log
tail -n 15 $OUT/pytest_queries.log
timeout 600 python tools/v6_probe.py --steps ${2:-300} --rounds ${3:-5} > $OUT/v6_probe.txt 2>&1
echo "probe exit: $?" >> $OUT/env.log
cat $OUT/env.log
cat $OUT/v6_probe.txt
