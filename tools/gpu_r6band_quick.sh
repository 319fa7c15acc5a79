#!/bin/bash
set -u
TAG=${1:-r6bandq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/rank_band_bench.py 20 2>&1 | grep -v amdgpu.ids | tee $OUT/rank_band_bench.txt
