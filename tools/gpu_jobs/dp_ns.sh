#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 200 python tools/dp_scale.py 1080p 2>&1 | grep "B=" | tee gpurun_out/dp_scale_1080p.txt
timeout -k 10 200 python tools/dp_scale.py teddy 2>&1 | grep "B=" | tee gpurun_out/dp_scale_teddy.txt
for sh in 540x960x512:1,2,4,8 1080x1920x512:1,2 1030x1040x1024:1,2,4 375x450x64:1,4,16,64 720x1280x192:1,2,4; do
  echo "== $sh"; timeout -k 10 200 python tools/dp_scale.py $sh 2>&1 | grep "B="
done | tee gpurun_out/dp_scale_other.txt
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "dp or config3" 2>&1 | tail -2
