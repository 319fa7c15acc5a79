#!/bin/bash
# round 5, second lease: the new tests, the one-call probe, the matrix-pipe probe by operand data
set -u
OUT=gpurun_out/r5b
mkdir -p $OUT
export TMPDIR=/tmp
export KGE_PLUGIN_LOG=$OUT/libkge_plugin_gpu.jsonl
timeout 900 python -m pytest tests/test_gpu_queries.py tests/test_gpu_libkge_plugin.py -m gpu -q --timeout=600 -p no:cacheprovider \
  -k "one_call_entry or test_j or test_k or group or out_of_memory" > $OUT/pytest_new.log 2>&1
echo "pytest new exit: $?" > $OUT/env.log
tail -n 15 $OUT/pytest_new.log | grep -v Warning
timeout 300 python tools/one_call_v8_probe.py > $OUT/one_call_v8_probe.txt 2>&1
cat $OUT/one_call_v8_probe.txt
timeout 120 python - > $OUT/mfma_by_operands.txt 2>&1 <<'PY'
import json, torch, bench
print(json.dumps(bench.matrix_pipe_probe(torch.device("cuda", 0))))
PY
cat $OUT/mfma_by_operands.txt
cat $OUT/env.log
