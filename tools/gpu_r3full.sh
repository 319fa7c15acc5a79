#!/bin/bash
# Round-3 full pass (runs ON the GPU box): all -m gpu tests, smoke, bench, rocprofv3 kernel trace of the bench,
# FETCH/WRITE counter passes.   bash tools/gpu_r3full.sh <tag> [nopmc]
set -u
TAG=${1:-r3}
MODE=${2:-full}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch;print(torch.cuda.get_device_name(0), torch.version.hip)" > $OUT/env.log 2>&1
KGE_BSHAPE_LOG=$R/$OUT/bshape_ranks.jsonl KGE_PLUGIN_LOG=$R/$OUT/plugin_gpu.jsonl timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $OUT/pytest_all.log 2>&1
echo "pytest all exit: $?" >> $OUT/env.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
echo "smoke exit: $?" >> $OUT/env.log
timeout 900 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" >> $OUT/env.log
cd /tmp
B="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-one-sided --streams 1"  # one launch at a time: the trace's average duration is roofline.avg_launch_us
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o bench -- $B > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err
echo "rocprof two-sided exit: $?" >> $R/$OUT/env.log
if [ "$MODE" != "nopmc" ]; then
for C in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $C -d $R/$OUT/pmc_$C -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-sided --streams 1 > /dev/null 2> $R/$OUT/pmc_$C.err
echo "pmc $C exit: $?" >> $R/$OUT/env.log
done
fi
cd $R
python tools/db_summary.py $OUT > $OUT/summary.txt 2>&1
tail -n 8 $OUT/pytest_all.log
cat $OUT/env.log
head -c 3000 $OUT/summary.txt
cat $OUT/bench.json
tail -n 3 $OUT/bench.err $OUT/smoke.log
