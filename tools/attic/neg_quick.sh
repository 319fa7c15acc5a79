#!/bin/bash
# quick iteration on the gather kernels: spo/neg parity tests + kernel trace of tools/neg_pmc.py
TAG=${1:-nq}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "golden or random_shapes or duplicates or emb_equals" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/neg_trace -o r -- python $GRAFT_REPO_ROOT/tools/neg_pmc.py > $OUT/neg_trace.out 2> $OUT/neg_trace.err
python - <<PY
import sqlite3
c = sqlite3.connect("$OUT/neg_trace/r_results.db")
for r in c.execute("select * from top_kernels limit 6"): print(r[0][:60], r[1], round(r[3],1))
PY
