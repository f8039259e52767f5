cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r03a}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1 ; echo "tests rc=$?" >> gpurun_out/${T}_tests.log
tail -3 gpurun_out/${T}_tests.log
PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout -k 10 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_nocache_tests.log 2>&1 ; echo "tests rc=$? (PYTORCH_NO_CUDA_MEMORY_CACHING=1: every tensor in its own mapping)" >> gpurun_out/${T}_nocache_tests.log
tail -3 gpurun_out/${T}_nocache_tests.log
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --shared-device --steps 3 --warmup 1 --batch 32 --also-batch 2 --also-steps 2 > gpurun_out/${T}_bench_gloo2.json 2> gpurun_out/${T}_bench_gloo2.err; echo "gloo2 rc=$?"
timeout -k 10 600 python bench.py --rehearse-nccl --steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-latency > gpurun_out/${T}_bench_nccl1.json 2> gpurun_out/${T}_bench_nccl1.err; echo "nccl1 rc=$?"
python3 -c "
import json
for f in ('gloo2','nccl1'):
    d=json.load(open('gpurun_out/${T}_bench_%s.json'%f)); print(f, d['n_gpus'], d['value'], d['sgm_status'], d['also']['value'], d['roofline']['form'])"
bash tools/profile_bench.sh $T > gpurun_out/${T}_profile.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/${T}_profile.log
