cd $GRAFT_REPO_ROOT
timeout 800 python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu 2>&1 | grep -v Warning | grep "AssertionError\|passed\|failed" | cut -c1-300 | head -40
