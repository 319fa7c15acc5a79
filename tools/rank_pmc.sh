#!/bin/bash
# Counter passes over the counting kernel (pairs_bf16_v4_kernel<.., V3_RANK>) at a Wikidata5M shard: matrix-pipe busy
# cycles and LDS activity, each --pmc set in its own run, no trace domains.  -> gpurun_out/r2_rank_pmc.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/rankpmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $OUT/avail_lds.txt
run() { timeout 50 rocprofv3 --pmc $2 -d $OUT/$1 -o r -- python $R/tools/rank_pmc.py > $OUT/$1.out 2> $OUT/$1.err; echo "$1 exit $?" >> $OUT/env.log; }
run a "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
run b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
cd $R
python - <<'PY' > gpurun_out/r2_rank_pmc.txt 2>&1
import glob, sqlite3, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/rankpmc"
print("rocprofv3 --pmc (separate passes) over tools/rank_pmc.py: 30 launches of pairs_bf16_v4_kernel<ComplEx, d=256, V3_RANK>, n=512 x 2 directions, E=574,311")
print("available LDS counters:", open(out + "/avail_lds.txt").read())
print(open(out + "/env.log").read())
for sub in ("a", "b"):
    for db in glob.glob(f"{out}/{sub}/**/*_results.db", recursive=True):
        con = sqlite3.connect(db)
        try:
            for r in con.execute("select counter_name, count(*), avg(value) from counters_collection "
                                 "where kernel_name like '%pairs_bf16_v4_kernel%' group by counter_name"):
                print(sub, r[0], "dispatches=%d mean/dispatch=%.1f" % (r[1], r[2]))
        except Exception as e:
            print(sub, "query failed:", e)
PY
cat gpurun_out/r2_rank_pmc.txt
