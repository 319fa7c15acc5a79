#!/bin/bash
# quick kernel iteration: queries tests + shared-kernel parity tests + bench   bash tools/gpu_r3b.sh <tag>
set -u
TAG=${1:-r3b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_queries.py -m gpu -q -x --timeout=600 -s > $OUT/pytest_queries.log 2>&1
echo "pytest queries exit: $?" > $OUT/env.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_score_rank.py tests/test_gpu_ce.py tests/test_gpu_fuzz_shapes.py -m gpu -q --timeout=600 > $OUT/pytest_parity.log 2>&1
echo "pytest parity exit: $?" >> $OUT/env.log
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" >> $OUT/env.log
tail -n 3 $OUT/pytest_queries.log
tail -n 3 $OUT/pytest_parity.log
cat $OUT/env.log
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
r = d["roofline"]
print("value %.3e ms_per_step %.4f host %.4f regions %s" % (d["value"], d["ms_per_step"], d["host_issue_ms_per_step"], [round(x, 4) for x in d["regions_ms_per_step"]]))
print("two-sided %.2f us frac %.3f | contiguous %.2f us %.3f | one-call %.2f us %.3f" % (r["avg_launch_us"], r["frac"], r["contiguous_pitch"]["avg_launch_us"], r["contiguous_pitch"]["frac"], r["one_call_entry_us"], r["one_call_entry_frac"]))
o = r["one_sided_launch"]
print("one-sided %.2f us frac %.3f | one-call %.2f us %.3f" % (o["avg_launch_us"], o["frac"], o["one_call_entry_us"], o["one_call_entry_frac"]))
print("split step %.2f us %.3f" % (r["split_query_step"]["avg_launch_us"], r["split_query_step"]["frac"]))
print({k: (round(v["avg_launch_us"], 1), round(v["frac"], 3)) for k, v in r["one_sided_by_batch"].items()})
print("rank", {k: (round(v["fused_us"], 1), round(v["frac"], 3)) for k, v in d["roofline_rank"].items() if isinstance(v, dict)})
PY
tail -n 5 $OUT/bench.err
