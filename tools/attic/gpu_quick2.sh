#!/bin/bash
# quick iteration on the headline kernel: bf16 parity tests, phase stamps, bench, one-/two-sided rocprofv3 averages
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -k "bf16 or emb_equals or empty or c2_full" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python tools/v2_phases.py > $OUT/v2_phases.txt 2>&1
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-one-sided > /dev/null 2> $R/$OUT/prof.err
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof1 -o one -- python $R/tools/one_sided.py > $R/$OUT/prof_one.txt 2> $R/$OUT/prof1.err
cd $R
python tools/db_summary.py $OUT | tee $OUT/summary.txt
grep -o '"avg_launch_us": [0-9.]*' $OUT/bench.json
grep -A3 "v4 (loader/consumer waves) n=512" $OUT/v2_phases.txt | head -3
