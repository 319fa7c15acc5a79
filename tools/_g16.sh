cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_bwd_gemm16.py -q -m gpu 2>&1 | tail -5
timeout 200 python tools/gemm16_probe.py 2>&1 | tail -14
