cd $GRAFT_REPO_ROOT
timeout 800 python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu -k "entity_ranking" 2>&1 | grep -v Warning | grep "Error\|passed\|failed" | cut -c1-300 | head -30
