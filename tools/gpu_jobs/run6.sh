cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r3e}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1 ; echo "tests rc=$?" >> gpurun_out/${T}_tests.log
tail -4 gpurun_out/${T}_tests.log
AB_SIZES="375,450,128,256;1080,1920,256,8" timeout -k 10 600 python tools/ab_sgm.py tools/_probe/libstm_B.so tools/_probe/libstm_C.so > gpurun_out/${T}_ab.log 2>&1; cat gpurun_out/${T}_ab.log
timeout -k 10 300 python tools/tile_stamps.py 375 450 128 64 > gpurun_out/${T}_stamps_teddy.log 2>&1; tail -10 gpurun_out/${T}_stamps_teddy.log
