cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r3h}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1 ; echo "tests rc=$?" >> gpurun_out/${T}_tests.log
tail -4 gpurun_out/${T}_tests.log
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"; python3 -c "
import json;d=json.load(open('gpurun_out/${T}_bench.json'));print(d['value'],d['stage_ms'],d['roofline']['frac']);print(d['also']['value'],d['also']['stage_ms'],d['also']['roofline']['frac']);print(d['latency_b1']);print(d['also_config3'])"
