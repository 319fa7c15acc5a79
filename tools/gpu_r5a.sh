#!/bin/bash
# round 5, first lease: the whole -m gpu suite (no -x: see every failure), smoke, a default bench line
set -u
OUT=gpurun_out/${1:-r5a}
mkdir -p $OUT
export TMPDIR=/tmp
export KGE_PLUGIN_LOG=$OUT/libkge_plugin_gpu.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest gpu exit: $?" > $OUT/env.log
tail -n 40 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit: $?" >> $OUT/env.log
tail -n 2 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" >> $OUT/env.log
cat $OUT/env.log
python - ${1:-r5a} <<'PY'
import json
import sys; d = json.load(open(f"gpurun_out/{sys.argv[1]}/bench.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"])
PY
