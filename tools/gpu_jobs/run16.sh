cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sgm_tiled.py tests/test_gpu_sgm_robust.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -6
for S in 1 2 3 4; do
  echo "== segs $S"; STM_TILE_SEGS=$S AB_SIZES="1080,1920,256,8" timeout -k 10 300 python tools/ab_sgm.py tools/_probe/libstm_S.so 2>&1 | grep -v amdgpu
done
echo "== walked (no rows)"; STM_TILE_NO_ROWS=1 AB_SIZES="1080,1920,256,8;1080,1920,256,4" timeout -k 10 300 python tools/ab_sgm.py tools/_probe/libstm_S.so 2>&1 | grep -v amdgpu
echo "== segs 2, 4 pairs"; AB_SIZES="1080,1920,256,4;1080,1920,256,3;1080,1920,256,16" timeout -k 10 300 python tools/ab_sgm.py tools/_probe/libstm_S.so 2>&1 | grep -v amdgpu
