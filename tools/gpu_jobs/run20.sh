cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_sgm_robust.py -x -q -k "bench_starts" 2>&1 | tail -5
