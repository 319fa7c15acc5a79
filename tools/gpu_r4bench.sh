#!/bin/bash
# bench line + smoke + the sharded tests on one lease   bash tools/gpu_r4bench.sh <tag> [steps] [warmup]
set -u
TAG=${1:-r4bench}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo > $OUT/env.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit: $?" >> $OUT/env.log
tail -n 3 $OUT/smoke.log
timeout 1500 python bench.py --gpus 1 --steps ${2:-20} --warmup ${3:-5} > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" >> $OUT/env.log
tail -n 5 $OUT/bench.err
timeout 900 env KGE_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --shape fb15k > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err
echo "bench dist1 exit: $?" >> $OUT/env.log
tail -n 5 $OUT/bench_dist1.err
cat $OUT/env.log
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "host_issue_ms_per_step")}))
print("roofline", json.dumps(d["roofline"]))
print("training", json.dumps(d["training_tolerance"]))
for k in ("one_launch_per_batch", "score_sp_groups", "score_sp_groups_by_batch", "roofline_train", "roofline_rank"):
    print(k, json.dumps(d.get(k)))
PY
tail -c 1500 $OUT/bench_dist1.json
