#!/bin/bash
# One gpurun call: parity tests, smoke, bench, per-kernel probe, rocprofv3 kernel trace.
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh [tag]
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch;print(torch.cuda.get_device_name(0), torch.version.hip)" > $OUT/env.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 > $OUT/pytest_all.log 2>&1
echo "pytest all exit: $?" >> $OUT/env.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
echo "smoke exit: $?" >> $OUT/env.log
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" >> $OUT/env.log
KGE_BENCH_FORCE_DIST=1 KGE_BENCH_EXCHANGE_GRAPH=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err
echo "dist(1 rank, RCCL) bench exit: $?" >> $OUT/env.log
timeout 300 python tools/v2_phases.py > $OUT/v2_phases.txt 2>&1
timeout 900 python tools/perf_probe.py > $OUT/probe.jsonl 2> $OUT/probe.err
echo "probe exit: $?" >> $OUT/env.log
# job-level probes: backward, negative sampling, one training step, evaluation, listed subsets
for P in bwd_probe ns_probe train_probe eval_probe subset_probe ce_probe ce_phases ce_host; do
  timeout 300 python tools/$P.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|_warn_once\|ROCTracer" > $OUT/$P.txt
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-one-sided > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err )
echo "rocprof exit: $?" >> $OUT/env.log
for C in FETCH_SIZE WRITE_SIZE; do
( cd /tmp && timeout 300 rocprofv3 --pmc $C -d $GRAFT_REPO_ROOT/$OUT/pmc_$C -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-sided > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$C.err )
echo "pmc $C exit: $?" >> $OUT/env.log
done
find $OUT/prof -name "*stats*" | head >> $OUT/env.log
# MFMA-utilisation counters of the headline kernel + HBM counters of the gather kernels
bash tools/gpu_pmc.sh ${TAG}pmc > $OUT/gpu_pmc.log 2>&1
tail -5 $OUT/pytest_all.log
cat $OUT/env.log
cat $OUT/bench.json
