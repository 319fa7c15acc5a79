#!/bin/bash
# iteration on the evaluation path: ranking tests (golden EntityRanking fixtures, rank-count parity) + probe
TAG=${1:-ev}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model_eval.py tests/test_gpu_parity.py -m gpu -q -x -k "rank or eval or ranking or filter_lookup" --timeout=300 > $OUT/pytest_eval.log 2>&1; tail -8 $OUT/pytest_eval.log
timeout 300 python tools/eval_probe.py 2>&1 | grep -v "amdgpu.ids" > $OUT/eval_probe.txt; head -30 $OUT/eval_probe.txt
