cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BS=2048 python $R/tools/eval_probe.py 2>&1 | grep -A22 "index build" | cut -c1-150
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/evf2 -o sk -- env BS=2048 python $R/tools/eval_probe.py > /dev/null 2>&1
python $R/tools/db_summary.py $R/gpurun_out/evf2 2>/dev/null | head -8 | cut -c1-170
