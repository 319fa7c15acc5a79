#!/bin/bash
# One gpurun call: counter passes only (each --pmc run on its own, --kernel-trace on its own).
#   bench.py        SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES / GRBM_GUI_ACTIVE  -> MFMA utilisation
#   tools/neg_pmc.py  kernel trace, FETCH_SIZE, WRITE_SIZE                          -> HBM GB/s (gather)
# Usage: bash tools/gpu_pmc.sh [tag];  then  python tools/pmc_summary.py --extra gpurun_out/<tag> <tag>
set -u
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY[A-Z_]*\|GRBM_GUI_ACTIVE\|SQ_WAVE_CYCLES\|SQ_CYCLES" | sort -u > $OUT/avail.txt
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-sided"
N="python $GRAFT_REPO_ROOT/tools/neg_pmc.py"
run() { # name, counters, cmd
  timeout 300 rocprofv3 --pmc $2 -d $OUT/$1 -o r -- $3 > $OUT/$1.out 2> $OUT/$1.err
  echo "$1 exit $?" >> $OUT/env.log
}
run mfma_a "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "$B"
run mfma_b "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "$B"
run mfma_c "GRBM_GUI_ACTIVE" "$B"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/neg_trace -o r -- $N > $OUT/neg_trace.out 2> $OUT/neg_trace.err
echo "neg_trace exit $?" >> $OUT/env.log
run neg_FETCH_SIZE FETCH_SIZE "$N"
run neg_WRITE_SIZE WRITE_SIZE "$N"
cat $OUT/env.log; cat $OUT/avail.txt
