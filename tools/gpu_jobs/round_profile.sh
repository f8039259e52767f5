cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r03b}
bash tools/profile_bench.sh $T > gpurun_out/${T}_profile.log 2>&1; echo "profile rc=$?"; tail -2 gpurun_out/${T}_profile.log
timeout -k 10 600 python tools/stage_times.py > gpurun_out/${T}_all_ops_stage_times.txt 2>&1; tail -40 gpurun_out/${T}_all_ops_stage_times.txt
PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout -k 10 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_nocache_tests.log 2>&1 ; echo "tests rc=$? (PYTORCH_NO_CUDA_MEMORY_CACHING=1: every tensor in its own mapping)" >> gpurun_out/${T}_nocache_tests.log
tail -3 gpurun_out/${T}_nocache_tests.log
timeout -k 10 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
