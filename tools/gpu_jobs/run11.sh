cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AB_OP=sgm AB_SIZES="375,450,128,256" timeout -k 10 600 python tools/ab_sgm.py tools/_probe/libstm_B.so tools/_probe/libstm_D.so tools/_probe/libstm_E.so 2>&1 | grep -v amdgpu
AB_SIZES="375,450,128,256;1080,1920,256,8" timeout -k 10 600 python tools/ab_sgm.py tools/_probe/libstm_B.so tools/_probe/libstm_D.so tools/_probe/libstm_E.so 2>&1 | grep -v amdgpu
timeout -k 10 300 python -m pytest tests/test_gpu_sgm_robust.py -x -q 2>&1 | tail -2
