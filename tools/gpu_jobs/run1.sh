set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_tests.log 2>&1 ; echo "tests rc=$?" >> gpurun_out/r3a_tests.log
tail -5 gpurun_out/r3a_tests.log
for mode in coop plain; do
  if [ $mode = plain ]; then export STM_SWEEP_PLAIN_LAUNCH=1; else unset STM_SWEEP_PLAIN_LAUNCH; fi
  echo "== $mode" >> gpurun_out/r3a_forms.log
  timeout -k 10 300 python tools/sgm_forms.py 375 450 128 256 >> gpurun_out/r3a_forms.log 2>&1
  timeout -k 10 300 python tools/sgm_forms.py 1080 1920 256 8 >> gpurun_out/r3a_forms.log 2>&1
done
unset STM_SWEEP_PLAIN_LAUNCH
cat gpurun_out/r3a_forms.log
timeout -k 10 300 python tools/tile_stamps.py 375 450 128 64 > gpurun_out/r3a_stamps_teddy.log 2>&1; tail -20 gpurun_out/r3a_stamps_teddy.log
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --shared-device --steps 3 --warmup 1 --batch 32 --also-batch 2 --also-steps 2 > gpurun_out/r3a_bench2.log 2>&1; echo "rc=$?" >> gpurun_out/r3a_bench2.log; tail -3 gpurun_out/r3a_bench2.log
timeout -k 10 600 python bench.py > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; echo "bench rc=$?"; cat gpurun_out/r3a_bench.json
