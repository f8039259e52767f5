cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sgm_tiled.py -x -q -k "128" 2>&1 | tail -15
STM_TILE_H2=1 AB_SIZES="375,450,128,256;375,450,128,1" timeout -k 10 300 python tools/ab_sgm.py tools/_probe/libstm_H4.so 2>&1 | grep -v amdgpu
AB_SIZES="375,450,128,256;375,450,128,1" timeout -k 10 300 python tools/ab_sgm.py tools/_probe/libstm_H4.so 2>&1 | grep -v amdgpu
