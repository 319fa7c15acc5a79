#!/bin/bash
# the paired KvsAll backward: its tests, the plugin's KvsAll cases, the bench leg   bash tools/gpu_r5kvs.sh <tag>
set -u
TAG=${1:-r5kvs}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ce.py tests/test_gpu_libkge_plugin.py tests/test_gpu_fuzz_shapes.py tests/test_gpu_sharded_train.py \
    -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit: $?" | tee $OUT/env.log
tail -n 12 $OUT/pytest.log | grep -v Warning
timeout 300 python -c "
import json, torch, bench
print(json.dumps(bench.kvsall_step_leg(torch.device('cuda', 0), 512, 50)))
" 2>&1 | grep -v amdgpu.ids | tee $OUT/kvsall_leg.json
