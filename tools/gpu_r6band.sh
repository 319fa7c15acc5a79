#!/bin/bash
# round 6: band-and-rescore -- the rank tests, then the three forms of the counting kernel timed
#   bash tools/gpu_r6band.sh <tag> [pytest -k expression]
set -u
TAG=${1:-r6band}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_score_rank.py -m gpu -q --timeout=900 -p no:cacheprovider -k "${2:-band}" > $OUT/pytest.log 2>&1
echo "pytest exit: $?" | tee $OUT/env.log
grep -E "^E  |passed|failed" $OUT/pytest.log | cut -c1-400 | head -40
timeout 600 python tools/rank_band_bench.py 20 2>&1 | grep -v amdgpu.ids | tee $OUT/rank_band_bench.txt
