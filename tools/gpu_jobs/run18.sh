cd $GRAFT_REPO_ROOT
AB_SIZES="1080,1920,256,8;1080,1920,256,1;540,960,256,16" timeout -k 10 600 python tools/ab_sgm.py tools/_probe/libstm_S2.so tools/_probe/libstm_E23.so 2>&1 | grep -v amdgpu
timeout -k 10 600 python -m pytest tests/test_gpu_sgm_tiled.py -x -q 2>&1 | tail -2
