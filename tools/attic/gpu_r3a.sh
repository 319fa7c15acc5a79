#!/bin/bash
# Round-3 iteration pass (runs ON the GPU box): consumer-loop microbenchmark, prepared / split query tests,
# the bf16 parity tests of the shared kernel, timing probe + phase stamps.   bash tools/gpu_r3a.sh <tag>
set -u
TAG=${1:-r3a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch;print(torch.cuda.get_device_name(0), torch.version.hip)" > $OUT/env.log 2>&1
timeout 120 ./tools/ubench/consumer_loop > $OUT/consumer_loop.txt 2>&1
echo "ubench exit: $?" >> $OUT/env.log
timeout 900 python -m pytest tests/test_gpu_queries.py -m gpu -q -x --timeout=600 -s > $OUT/pytest_queries.log 2>&1
echo "pytest queries exit: $?" >> $OUT/env.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_score_rank.py -m gpu -q --timeout=600 > $OUT/pytest_parity.log 2>&1
echo "pytest parity exit: $?" >> $OUT/env.log
timeout 600 python tools/prep_probe.py > $OUT/prep_probe.txt 2> $OUT/prep_probe.err
echo "probe exit: $?" >> $OUT/env.log
tail -3 $OUT/pytest_queries.log $OUT/pytest_parity.log
cat $OUT/env.log $OUT/consumer_loop.txt
grep -v "^$" $OUT/prep_probe.txt | tail -30
tail -5 $OUT/prep_probe.err
