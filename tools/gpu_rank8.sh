#!/bin/bash
# iteration on the counting kernel   bash tools/gpu_rank8.sh <tag>
set -u
TAG=${1:-rank8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo > $OUT/env.log
if [ "${RANK8_SKIP_TESTS:-0}" != 1 ]; then
timeout 1200 python -m pytest tests/test_gpu_score_rank.py -m gpu -q -x --timeout=900 > $OUT/pytest_rank.log 2>&1
echo "pytest rank exit: $?" >> $OUT/env.log
tail -n 30 $OUT/pytest_rank.log
fi
if [ "${RANK8_SKIP_TESTS:-0}" = 1 ]; then echo skipped tests; fi
timeout 600 python tools/rank8_probe.py > $OUT/rank8_probe.txt 2>&1
echo "probe exit: $?" >> $OUT/env.log
if [ "${RANK8_PROF:-0}" = 1 ]; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o rank8 -- python $GRAFT_REPO_ROOT/tools/rank8_probe.py > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
  echo "rocprof exit: $?" >> $OUT/env.log
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(r.get("Name", "")[:90], r.get("Calls"), r.get("AverageNs"), r.get("Percentage"))
PY
fi
cat $OUT/env.log
cat $OUT/rank8_probe.txt
