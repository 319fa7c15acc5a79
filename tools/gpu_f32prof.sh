#!/bin/bash
# kernel split of the float32-scoring 1vsAll step (the default LibKGE configuration's path)   bash tools/gpu_f32prof.sh <tag>
set -u
TAG=${1:-f32prof}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ONLY=f32_scoring STEPS=60 GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o train -- python $GRAFT_REPO_ROOT/tools/train_step_prof.py > $OUT/prof.log 2>&1
echo "rocprof exit: $?"
tail -3 $OUT/prof.log
