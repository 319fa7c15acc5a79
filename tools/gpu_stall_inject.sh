#!/bin/bash
# The race / timing pass over the hand-synchronised kernels (SURVEY.md 5 "Race detection"; VERDICT r5 missing 6):
# rebuild libkge_amd.so ON THE GPU BOX with -DKGE_STALL_INJECT (random sleeps in front of every workgroup barrier and
# LDS-DMA piece: common.hpp kge_stall), then run the bit-equality tests -- every matrix-core kernel against the oracle,
# against its sibling kernels and against the two-step paths -- REPS times under it.  The box is scratch: the product
# build in the repository is not touched.      bash tools/gpu_stall_inject.sh <tag> [reps]
set -u
TAG=${1:-r6stall}
REPS=${2:-3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/kge_amd/csrc
make clean > /dev/null
( time make -j"$(nproc)" CXXEXTRA=-DKGE_STALL_INJECT ) > $OUT/build.log 2>&1
echo "build exit: $?" | tee $OUT/summary.txt
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee -a $OUT/summary.txt
import ctypes
from kge_amd import _lib
L = _lib.lib()
L.kge_debug_stall_build.restype = ctypes.c_int
print("kge_debug_stall_build() =", L.kge_debug_stall_build())
assert L.kge_debug_stall_build() == 1
PY
TESTS="tests/test_gpu_queries.py tests/test_gpu_score_rank.py tests/test_gpu_ce.py tests/test_gpu_bwd_gemm16.py tests/test_gpu_parity.py tests/test_gpu_fuzz_shapes.py tests/test_gpu_stress_streams.py tests/test_gpu_train_graph.py"
for r in $(seq 1 $REPS); do
  timeout 2400 python -m pytest $TESTS -m gpu -q --timeout=1200 -p no:cacheprovider -p no:randomly > $OUT/pytest_rep$r.log 2>&1
  echo "rep $r: pytest exit $? -- $(tail -n 1 $OUT/pytest_rep$r.log)" | tee -a $OUT/summary.txt
done
python - <<'PY' | tee -a $OUT/summary.txt
import ctypes
from kge_amd import _lib
L = _lib.lib()
try:
    L.kge_debug_launch_count.restype = ctypes.c_int
    L.kge_debug_launch_count.argtypes = [ctypes.c_int]
except Exception:
    pass
PY
grep -E "^FAILED|^ERROR" $OUT/pytest_rep*.log | head -20 | tee -a $OUT/summary.txt
