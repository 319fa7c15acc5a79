#!/bin/bash
# iteration on the fused 1vsAll loss: its tests + the v3 store-path tests (same kernel template) + timing
TAG=${1:-ce}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ce.py -m gpu -q -x --timeout=300 > $OUT/pytest_ce.log 2>&1; tail -15 $OUT/pytest_ce.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bf16" --timeout=300 > $OUT/pytest_bf16.log 2>&1; tail -3 $OUT/pytest_bf16.log
timeout 120 python tools/ce_phases.py 2>&1 | grep -v "amdgpu.ids" > $OUT/ce_phases.txt; cat $OUT/ce_phases.txt
timeout 300 python tools/ce_probe.py > $OUT/ce_probe.txt 2>&1; cat $OUT/ce_probe.txt | grep -v "amdgpu.ids\|UserWarning\|_warn_once"
timeout 120 python tools/ce_host.py 2>&1 | grep -v "amdgpu.ids" > $OUT/ce_host.txt; cat $OUT/ce_host.txt
