cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r3f}
timeout -k 10 900 python -m pytest tests/test_gpu_sgm_tiled.py tests/test_gpu_sgm_robust.py -x -q > gpurun_out/${T}_tests.log 2>&1 ; echo "tests rc=$?" >> gpurun_out/${T}_tests.log
tail -4 gpurun_out/${T}_tests.log
for shape in "375 450 128 1" "375 450 128 4" "375 450 128 16" "1080 1920 256 1" "1080 1920 256 2" "540 960 128 1" "540 960 256 1"; do
  echo "== $shape" >> gpurun_out/${T}_forms.log
  timeout -k 10 300 python tools/sgm_forms.py $shape 10 >> gpurun_out/${T}_forms.log 2>&1
done
grep -v amdgpu.ids gpurun_out/${T}_forms.log
