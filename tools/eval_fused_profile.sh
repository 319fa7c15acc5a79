#!/bin/bash
# Evaluation with the counts taken inside the scoring kernel (kge_score_rank_sp_po) against the two-step path:
# evaluator wall time, per-kernel split, single-batch timings up to the Wikidata5M-shard shape.
#   gpurun -- 'bash tools/eval_fused_profile.sh'   ->  gpurun_out/r2_eval_fused.txt  (copy to profiles/)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_eval_fused.txt
mkdir -p $R/gpurun_out
{
echo "# tools/eval_probe.py (C4 shape: E=14,541, d=512 DistMult bf16, 17,535 valid triples; raw + filtered + filtered-with-test, both directions)"
echo "## counting inside the scoring kernel (kge_score_rank_sp_po), full batches replayed as one hipGraph (default; \"second\" = the run that reuses the captured graph)"
python $R/tools/eval_probe.py 2>&1 | grep "index build"
BS=2048 python $R/tools/eval_probe.py 2>&1 | grep "index build"
BS=128 python $R/tools/eval_probe.py 2>&1 | grep "index build"
echo "## the same loop issued launch by launch (KGE_EVAL_GRAPH=0: no hipGraph replay of the full batches)"
KGE_EVAL_GRAPH=0 python $R/tools/eval_probe.py 2>&1 | grep "index build"
KGE_EVAL_GRAPH=0 BS=2048 python $R/tools/eval_probe.py 2>&1 | grep "index build"
KGE_EVAL_GRAPH=0 BS=128 python $R/tools/eval_probe.py 2>&1 | grep "index build"
echo "## two-step (KGE_EVAL_TWO_STEP=1: kge_score_sp_po + kge_rank_counts_multi)"
KGE_EVAL_TWO_STEP=1 python $R/tools/eval_probe.py 2>&1 | grep "index build"
KGE_EVAL_TWO_STEP=1 BS=2048 python $R/tools/eval_probe.py 2>&1 | grep "index build"
KGE_EVAL_TWO_STEP=1 BS=128 python $R/tools/eval_probe.py 2>&1 | grep "index build"
echo "## rocprofv3 --kernel-trace --stats over eval_probe.py, fused, batch 512 (3 evaluations of 35 batches)"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/evf -o sk -- python $R/tools/eval_probe.py > /dev/null 2>&1
python $R/tools/db_summary.py $R/gpurun_out/evf 2>/dev/null | head -10 | cut -c1-170
echo "## the same, batch 2048 (3 evaluations of 9 batches)"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/evf2 -o sk -- env BS=2048 python $R/tools/eval_probe.py > /dev/null 2>&1
python $R/tools/db_summary.py $R/gpurun_out/evf2 2>/dev/null | head -6 | cut -c1-170
echo "## tools/score_rank_probe.py: one batch, wall clock incl. host issue (two-step = score_sp_po + 2 rank_counts_multi; fused = kge_score_rank_sp_po), ComplEx"
for a in "512 14541 512" "2048 14541 512" "512 40943 512" "512 574311 256" "512 574311 512" "2048 574311 256"; do
  python $R/tools/score_rank_probe.py $a 2>&1 | grep "us per batch"
done
echo "## rocprofv3 --kernel-trace --stats over score_rank_probe.py 512 574311 256"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/srp2 -o sk -- python $R/tools/score_rank_probe.py 512 574311 256 > /dev/null 2>&1
python $R/tools/db_summary.py $R/gpurun_out/srp2 2>/dev/null | head -5 | cut -c1-170
} > $O 2>&1
cat $O
