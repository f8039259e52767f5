set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r3b}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1 ; echo "tests rc=$?" >> gpurun_out/${T}_tests.log
tail -5 gpurun_out/${T}_tests.log
timeout -k 10 300 python tools/sgm_forms.py 375 450 128 256 > gpurun_out/${T}_forms.log 2>&1
timeout -k 10 300 python tools/sgm_forms.py 1080 1920 256 8 >> gpurun_out/${T}_forms.log 2>&1
cat gpurun_out/${T}_forms.log
timeout -k 10 300 python tools/tile_stamps.py 375 450 128 64 > gpurun_out/${T}_stamps_teddy.log 2>&1; tail -12 gpurun_out/${T}_stamps_teddy.log
timeout -k 10 300 python tools/tile_stamps.py 1080 1920 256 8 > gpurun_out/${T}_stamps_1080.log 2>&1; tail -24 gpurun_out/${T}_stamps_1080.log
bash tools/time_libs.sh "prod" "375 450 128 256" "1080 1920 256 8" > /dev/null 2>&1; tail -14 gpurun_out/time_libs.log
