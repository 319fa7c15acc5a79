"""Recipe for `oracle/_ref/`: the reference package itself, placed beside the oracle so that it can run where
/root/reference does not exist (the GPU box).

TEST INFRASTRUCTURE ONLY (see oracle/kge_oracle.c): nothing under kge_amd/ reads oracle/_ref/.  It is used by
  * tests/test_gpu_libkge_plugin.py and the other plugin tests: an UNMODIFIED LibKGE drives the kernels on the MI355X;
  * bench.py's cpu_baseline leg: the reference's own KgeModel.score_sp / score_po timed on the GPU box's host cores
    (`cpu_baseline.kind` = "reference", BASELINE.md section 4: no re-implementation stands in for the reference).

The reference is pure Python: "compiling it from the sources where they lie" is a copy.  `__graft_entry__.build()` runs
this recipe whenever /root/reference is present (the build container); the output goes ONLY to oracle/_ref/, which is
git-ignored (reference sources never enter the history) but not gpurun-ignored, so it travels to the GPU box with the
snapshot like the built .so files.  On a box without /root/reference the recipe does nothing and whatever the snapshot
brought stays.

    python oracle/make_ref.py            # build container: (re)creates oracle/_ref/libkge/kge and .../tests/data
    python oracle/make_ref.py --clean    # removes oracle/_ref/libkge

What is copied: the `kge` package (*.py, *.yaml: 648 KB) and the two tiny datasets of the reference's own tests
(tests/data: 48 KB; the two-rank plugin job test trains on dataset_test).  oracle/ref_harness.py finds the copy
(`_find_root`) after KGE_REFERENCE_ROOT and /root/reference.
"""
import os
import shutil
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("KGE_REFERENCE_SRC", "/root/reference")
DST = os.path.join(_HERE, "_ref", "libkge")


def make(force: bool = False) -> bool:
    """True if oracle/_ref/libkge holds the package afterwards."""
    src_pkg = os.path.join(SRC, "kge")
    if not os.path.isdir(src_pkg):
        return os.path.isdir(os.path.join(DST, "kge"))
    stamp = os.path.join(DST, ".made_from")
    newest = 0.0
    for root, _dirs, files in os.walk(src_pkg):
        for f in files:
            if f.endswith((".py", ".yaml")):
                newest = max(newest, os.path.getmtime(os.path.join(root, f)))
    if not force and os.path.isfile(stamp) and os.path.getmtime(stamp) >= newest and os.path.isdir(os.path.join(DST, "kge")):
        return True
    shutil.rmtree(DST, ignore_errors=True)
    os.makedirs(DST)
    ignore = shutil.ignore_patterns("__pycache__", "*.pyc", "*.so", "*.pckl")
    shutil.copytree(src_pkg, os.path.join(DST, "kge"), ignore=ignore)
    data = os.path.join(SRC, "tests", "data")
    if os.path.isdir(data):
        shutil.copytree(data, os.path.join(DST, "tests", "data"), ignore=ignore)
    with open(stamp, "w") as f:
        f.write(SRC + "\n")
    return True


def clean() -> None:
    shutil.rmtree(DST, ignore_errors=True)


if __name__ == "__main__":
    if "--clean" in sys.argv:
        clean()
    else:
        print("oracle/_ref/libkge:", "present" if make(force="--force" in sys.argv) else "absent (no reference tree here)")
