set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r3d}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1 ; echo "tests rc=$?" >> gpurun_out/${T}_tests.log
tail -5 gpurun_out/${T}_tests.log
rm -f gpurun_out/time_libs.log
bash tools/time_libs.sh "A B" "375 450 128 256" "1080 1920 256 8" > /dev/null 2>&1
cp gpurun_out/time_libs.log gpurun_out/${T}_time.log; cat gpurun_out/${T}_time.log
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"; cat gpurun_out/${T}_bench.json
