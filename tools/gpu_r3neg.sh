#!/bin/bash
# rocprofv3 passes over the gather-bound negative-sampling kernel (runs ON the GPU box): kernel trace, FETCH_SIZE and
# WRITE_SIZE, each in a run of its own, for the WN18RR-shape table and for one beyond the Infinity Cache.
#   bash tools/gpu_r3neg.sh <tag>;  then  python tools/neg_pmc_summary.py gpurun_out/<tag> <tag>
set -u
TAG=${1:-r3neg}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
N="python $GRAFT_REPO_ROOT/tools/neg_pmc.py"
for CASE in wn18rr big; do
  export NEG_PMC_CASE=$CASE
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/neg_${CASE}_trace -o r -- $N > $OUT/neg_${CASE}_trace.out 2> $OUT/neg_${CASE}_trace.err
  echo "neg $CASE trace exit $?" >> $OUT/env.log
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C -d $OUT/neg_${CASE}_$C -o r -- $N > $OUT/neg_${CASE}_$C.out 2> $OUT/neg_${CASE}_$C.err
    echo "neg $CASE $C exit $?" >> $OUT/env.log
  done
done
cd $GRAFT_REPO_ROOT
cat $OUT/env.log
python tools/neg_pmc_summary.py gpurun_out/$TAG $TAG
