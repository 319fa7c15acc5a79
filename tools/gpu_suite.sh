#!/bin/bash
# the whole -m gpu suite + smoke on one lease   bash tools/gpu_suite.sh <tag> [extra pytest args]
set -u
TAG=${1:-suite}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export KGE_PLUGIN_LOG=$OUT/plugin.jsonl
timeout 3000 python -m pytest tests -m gpu -q --timeout=1200 ${2:-} > $OUT/pytest_gpu.log 2>&1
echo "pytest gpu exit: $?" > $OUT/env.log
tail -n 25 $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit: $?" >> $OUT/env.log
tail -n 3 $OUT/smoke.log
cat $OUT/env.log
