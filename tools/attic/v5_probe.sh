#!/bin/bash
# v5 (workgroup-local query build) against v4: bf16 parity tests, one-/two-sided launch times, phase stamps
export TMPDIR=/tmp
for V in 0 1; do
echo "=== KGE_V5=$V"
KGE_V5=$V timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -k "bf16 or emb_equals or empty or c2_full or c5" 2>&1 | tail -3
KGE_V5=$V python tools/one_sided.py --steps 200
KGE_V5=$V python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-one-sided --repeats 3 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('two-sided avg_launch_us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
done
KGE_V5=1 python tools/v2_phases.py 2>&1 | grep -A16 "n=512 workspace=True" | head -24
