#!/bin/bash
# DP tests with the forward pass forced to each wavefront form, then the timing table.
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for lanes in 16 64; do
  STM_DP_LANES=$lanes timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "dp or config3" > gpurun_out/dp_tests_$lanes.log 2>&1
  echo "lanes=$lanes rc=$?"; tail -3 gpurun_out/dp_tests_$lanes.log
done
timeout -k 10 300 python tools/dp_scale.py 1080p 2>&1 | tee gpurun_out/dp_scale_1080p.txt
timeout -k 10 200 python tools/dp_scale.py teddy 2>&1 | tee gpurun_out/dp_scale_teddy.txt
