cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r3i}
for shape in "375 450 128 2" "375 450 128 4" "375 450 128 16" "375 450 128 24" "375 450 128 32" "1080 1920 256 2" "1080 1920 256 3" "240 320 128 1" "96 128 128 1" "64 64 256 1"; do
  echo "== $shape"; timeout -k 10 300 python tools/sgm_forms.py $shape 10 2>&1 | grep -v "amdgpu.ids\|equal"
done > gpurun_out/${T}_forms.log 2>&1
cat gpurun_out/${T}_forms.log
