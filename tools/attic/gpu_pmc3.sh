#!/bin/bash
# round 3: FETCH_SIZE / WRITE_SIZE passes of the bench step (separate passes) -> gpurun_out/<tag>/pmc_*   bash tools/gpu_pmc3.sh <tag>
set -u
TAG=${1:-r3p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-one-sided > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $C -d $R/$OUT/pmc_$C -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-sided --streams 1 > /dev/null 2> $R/$OUT/pmc_$C.err
echo "pmc $C exit: $?"
done
cd $R
python tools/pmc_summary.py $OUT r3
