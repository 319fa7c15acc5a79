#!/bin/bash
# Where the split-query group launch (the headline's kernel, pairs_bf16_v8_kernel<ComplEx, SPLIT>) spends its wave
# cycles: SQ counter passes (each --pmc set in its own run, counters only) over tools/v8_pmc_target.py = 12 launches per
# query mode of the bench's group launch.  -> gpurun_out/<tag>/split_store_pmc.txt     bash tools/split_store_pmc.sh <tag>
set -u
TAG=${1:-r6splitpmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { timeout 120 rocprofv3 --pmc $2 -d $OUT/$1 -o v8 -- python $R/tools/v8_pmc_target.py > $OUT/$1.out 2> $OUT/$1.err; echo "$1 ($2) exit $?" >> $OUT/env.log; }
run a "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
run b "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
run c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
run d "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR"
run e "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
cd $R
python - $OUT <<'PY' | tee $OUT/split_store_pmc.txt
import glob, sqlite3, sys
out = sys.argv[1]
print("rocprofv3 --pmc (separate passes, counters only) over tools/v8_pmc_target.py: 12 group launches of 8 two-sided batches per query mode")
print("(n = 512, E = 14,541, d = 512); mean per dispatch; kernel <0, 1, ...> = split queries (the headline), <0, 0, ...> = single-pass")
print(open(out + "/env.log").read())
vals = {}
for sub in "abcde":
    for db in glob.glob(f"{out}/{sub}/**/*_results.db", recursive=True):
        con = sqlite3.connect(db)
        try:
            for r in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                 "where kernel_name like '%pairs_bf16_v8_kernel%' group by kernel_name, counter_name"):
                mode = "split" if "<0, 1," in r[0].replace("(int)", "") else "single"
                vals.setdefault(mode, {})[r[1]] = r[3]
                print(f"{mode:6s} {r[1]:28s} dispatches={r[2]:3d} mean/dispatch={r[3]:16.1f}")
        except Exception as e:
            print(sub, "query failed:", e)
for mode, v in vals.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CU_CYCLES" in v:
        print(f"{mode}: matrix-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) = {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * v['SQ_BUSY_CU_CYCLES']):.3f}")
    if "SQ_WAVE_CYCLES" in v:
        w = v["SQ_WAVE_CYCLES"]
        parts = {k: v[k] / w for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in v}
        print(f"{mode}: of the wave cycles: " + ", ".join(f"{k} {x:.3f}" for k, x in parts.items()))
    if "SQ_LDS_IDX_ACTIVE" in v and "SQ_BUSY_CU_CYCLES" in v:
        print(f"{mode}: LDS array active share = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES = {v['SQ_LDS_IDX_ACTIVE'] / v['SQ_BUSY_CU_CYCLES']:.3f}; bank conflict cycles {v.get('SQ_LDS_BANK_CONFLICT', float('nan')):.0f}")
PY
find $OUT -name "*.db" -delete 2>/dev/null
