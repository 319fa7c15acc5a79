#!/bin/bash
TAG=${1:-opt}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_optim.py tests/test_gpu_ce.py -m gpu -q -x --timeout=300 > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log
timeout 300 python tools/train_probe.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|_warn_once\|ROCTracer" > $OUT/train_probe.txt; head -2 $OUT/train_probe.txt | cut -c1-400; grep -n "one-pass" -A30 $OUT/train_probe.txt | grep "adagrad\|Self CUDA time" | cut -c1-200
