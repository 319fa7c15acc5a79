#!/bin/bash
# Round-2 GPU pass (runs ON the GPU box, from the repo root): parity tests incl. the LibKGE plugin
# (when the reference package was shipped, tools/gpu_plugin.sh), bench, rocprofv3 kernel traces and
# FETCH/WRITE counters of the two-sided and one-sided launches (+ padded pitch), phase stamps.
#   bash tools/gpu_r2.sh <tag> [quick|lite]   (lite: everything but the one-sided traces and the PMC passes)
set -u
TAG=${1:-r10}
MODE=${2:-full}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch;print(torch.cuda.get_device_name(0), torch.version.hip)" > $OUT/env.log 2>&1
if [ "$MODE" = "full" ] || [ "$MODE" = "lite" ]; then
KGE_BSHAPE_LOG=$R/$OUT/bshape_ranks.jsonl KGE_PLUGIN_LOG=$R/$OUT/plugin_gpu.jsonl timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > $OUT/pytest_all.log 2>&1
echo "pytest all exit: $?" >> $OUT/env.log
grep -h "PLUGIN_GPU\|BSHAPE_RANKS" $OUT/pytest_all.log > $OUT/plugin_lines.txt
else
KGE_PLUGIN_LOG=$R/$OUT/plugin_gpu.jsonl timeout 1500 python -m pytest tests/test_gpu_libkge_plugin.py tests/test_gpu_bshape_ranks.py tests/test_gpu_model_eval.py -m gpu -q --timeout=900 -rA > $OUT/pytest_new.log 2>&1
echo "pytest new exit: $?" >> $OUT/env.log
fi
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
echo "smoke exit: $?" >> $OUT/env.log
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" >> $OUT/env.log
if [ "$MODE" != "lite" ]; then
timeout 300 python tools/v2_phases.py > $OUT/v2_phases.txt 2>&1
timeout 200 python tools/gemm16_probe.py > $OUT/gemm16_probe.txt 2>&1
timeout 200 python tools/gemm16_phases.py > $OUT/gemm16_phases.txt 2>&1
fi
# the sharded step with ONE rank (RCCL init, exchange, padded slabs): the code path of bench.py --gpus N
for SH in wikidata5m fb15k; do
KGE_BENCH_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --shape $SH --no-cpu-baseline > $OUT/bench_dist1_$SH.json 2> $OUT/bench_dist1_$SH.err
echo "bench dist1 $SH exit: $?" >> $OUT/env.log
done
cd /tmp
B="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-one-sided"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o bench -- $B > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err
echo "rocprof two-sided exit: $?" >> $R/$OUT/env.log
if [ "$MODE" != "lite" ]; then
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof1 -o one -- python $R/tools/one_sided.py > $R/$OUT/prof_one.txt 2> $R/$OUT/prof1.err
echo "rocprof one-sided exit: $?" >> $R/$OUT/env.log
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof1p -o one -- python $R/tools/one_sided.py --pad > $R/$OUT/prof_one_pad.txt 2> $R/$OUT/prof1p.err
for C in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $C -d $R/$OUT/pmc_$C -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-sided > /dev/null 2> $R/$OUT/pmc_$C.err
echo "pmc $C exit: $?" >> $R/$OUT/env.log
timeout 300 rocprofv3 --pmc $C -d $R/$OUT/pmc1_$C -o one -- python $R/tools/one_sided.py --steps 20 > /dev/null 2> $R/$OUT/pmc1_$C.err
timeout 300 rocprofv3 --pmc $C -d $R/$OUT/pmc1p_$C -o one -- python $R/tools/one_sided.py --steps 20 --pad > /dev/null 2> $R/$OUT/pmc1p_$C.err
done
fi
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_step -o step -- python $R/tools/step_kernels.py > $R/$OUT/step_kernels.log 2>&1
cd $R
python tools/db_summary.py $OUT > $OUT/summary.txt 2>&1
tail -5 $OUT/pytest_*.log
cat $OUT/env.log $OUT/summary.txt
cat $OUT/bench.json
