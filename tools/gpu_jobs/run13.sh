cd $GRAFT_REPO_ROOT
AB_SIZES="1080,1920,256,8;540,960,256,1" timeout -k 10 600 python tools/ab_sgm.py tools/_probe/libstm_F.so tools/_probe/libstm_G.so 2>&1 | grep -v amdgpu
timeout -k 10 600 python -m pytest tests/test_gpu_sgm_tiled.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
