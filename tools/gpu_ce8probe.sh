#!/bin/bash
# bash tools/gpu_ce8probe.sh <tag>: per-kernel times of the fused-loss passes under rocprofv3, product build and the chains-alone probe build
set -u
TAG=${1:-ce8probe}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for V in 0 1; do
  if [ $V = 1 ]; then
    ( cd $GRAFT_REPO_ROOT/kge_amd/csrc && rm -f ce_pairs_v8.o && make CXXEXTRA=-DKGE_V8C_PROBE=1 > /dev/null 2>&1 )
  fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof$V -o ce -- python $GRAFT_REPO_ROOT/tools/ce8_probe.py > $OUT/run$V.txt 2>&1
  grep "CE_V8" $OUT/run$V.txt
  python $GRAFT_REPO_ROOT/tools/db_summary.py $OUT/prof$V | grep -i "ce_kernel\|v4_kernel\|query_build\|combine" 
done
