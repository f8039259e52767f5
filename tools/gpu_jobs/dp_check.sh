#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "dp or config3 or integration or pipeline" > gpurun_out/dp_check_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/dp_check_tests.log
timeout -k 10 300 python bench.py > gpurun_out/dp_check_bench.json 2> gpurun_out/dp_check_bench.err; echo "bench rc=$?"
timeout -k 10 200 python tools/stage_times.py 2>&1 | tail -30 > gpurun_out/dp_check_stage_times.txt
