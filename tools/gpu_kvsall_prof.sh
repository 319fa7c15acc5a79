#!/bin/bash
# kernel split of the KvsAll training step of bench.kvsall_step_leg   bash tools/gpu_kvsall_prof.sh <tag>
set -u
TAG=${1:-kvsprof}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kvs -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import torch, bench
print(bench.kvsall_step_leg(torch.device('cuda', 0), 512, 50))
" > $OUT/prof.log 2>&1
echo "rocprof exit: $?"
grep -v "^W20\|amdgpu.ids" $OUT/prof.log | tail -3
