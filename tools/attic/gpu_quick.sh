#!/bin/bash
# quick iteration: bf16 kernel tests + phase stamps + bench (no cpu baseline)
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bf16 or emb_equals or empty" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python tools/v2_phases.py > $OUT/v2_phases.txt 2>&1
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
grep -o '"avg_launch_us": [0-9.]*' $OUT/bench.json
