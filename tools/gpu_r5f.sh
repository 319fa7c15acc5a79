#!/bin/bash
# the split-query store kernel with 16-byte stores: correctness (group launches against single launches and the oracle,
# smoke) and the A/B timing
set -u
OUT=gpurun_out/${1:-r5f}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_queries.py tests/test_gpu_full_size.py tests/test_gpu_stress_streams.py -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit: $?" > $OUT/env.log
tail -n 4 $OUT/pytest.log | grep -v Warning
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit: $?" >> $OUT/env.log
timeout 300 python tools/split_var_probe.py x4=0 x4=1 2>&1 | grep -v amdgpu.ids | grep '"split": 1' > $OUT/split_x4_probe.txt
cat $OUT/split_x4_probe.txt
cat $OUT/env.log
