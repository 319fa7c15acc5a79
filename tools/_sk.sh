cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sk -o sk -- python $R/tools/step_kernels.py > $R/gpurun_out/sk.log 2>&1
python $R/tools/db_summary.py $R/gpurun_out/sk 2>/dev/null | head -40 || ls -R $R/gpurun_out/sk | head
