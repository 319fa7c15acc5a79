#!/bin/bash
# round 5, third lease: split kernel with 4 reads in flight (no spills), store policy for unaligned rows, sharded NS test
set -u
OUT=gpurun_out/r5c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_queries.py tests/test_gpu_libkge_plugin.py tests/test_gpu_full_size.py -m gpu -q --timeout=600 -p no:cacheprovider \
  -k "test_gpu_queries or test_j or test_gpu_full_size" > $OUT/pytest_new.log 2>&1
echo "pytest exit: $?" > $OUT/env.log
tail -n 6 $OUT/pytest_new.log | grep -v Warning
timeout 300 python tools/v8_probe.py --steps 80 > $OUT/v8_probe.txt 2>&1
grep "group of 8\|one batch" $OUT/v8_probe.txt
for pol in default 0 2; do
  if [ $pol = default ]; then unset KGE_V4_STORE_SC1; else export KGE_V4_STORE_SC1=$pol; fi
  echo "== store policy $pol" >> $OUT/one_call_v8_probe.txt
  timeout 300 python tools/one_call_v8_probe.py >> $OUT/one_call_v8_probe.txt 2>&1
done
unset KGE_V4_STORE_SC1
grep -v amdgpu.ids $OUT/one_call_v8_probe.txt | cut -c1-400
cat $OUT/env.log
