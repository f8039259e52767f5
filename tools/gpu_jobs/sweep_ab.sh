#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_sgm_tiled.py tests/test_gpu_sgm_robust.py -m gpu -x -q > gpurun_out/sweep_ab_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/sweep_ab_tests.log
AB_SIZES="${AB_SIZES:-375,450,128,256;375,450,128,64}" timeout -k 10 300 python tools/ab_sgm.py stereo-matching_amd/stereomatch/libstereomatch_hip.so tools/_probe/libstm_before.so 2>&1 | tail -12 | tee gpurun_out/sweep_ab.txt
