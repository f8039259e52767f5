cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r3g}
timeout -k 10 900 python -m pytest tests/test_gpu_sgm_tiled.py -x -q > gpurun_out/${T}_tests.log 2>&1 ; echo "tests rc=$?" >> gpurun_out/${T}_tests.log
tail -3 gpurun_out/${T}_tests.log
rm -f gpurun_out/time_libs.log
cd /tmp && export TMPDIR=/tmp
for SHAPE in "375 450 128 1" "1080 1920 256 1"; do
  for FORM in walks paired; do
    D=/tmp/tl_$(echo $SHAPE $FORM | tr ' ' _); rm -rf $D
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python3 $GRAFT_REPO_ROOT/tools/run_lib.py $GRAFT_REPO_ROOT/tools/_probe/libstm_W.so $SHAPE $FORM > $D.log 2>&1
    echo "== $SHAPE $FORM" >> $GRAFT_REPO_ROOT/gpurun_out/${T}_b1_kernels.log
    python3 - $D >> $GRAFT_REPO_ROOT/gpurun_out/${T}_b1_kernels.log <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sgm_" in r["Name"]:
            n = r["Name"].replace("void stm::(anonymous namespace)::", "").split("(")[0]
            print("  %-40s calls %4s avg_us %10.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/${T}_b1_kernels.log
for shape in "375 450 128 1" "375 450 128 8" "1080 1920 256 1"; do
  echo "== $shape"; timeout -k 10 300 python tools/sgm_forms.py $shape 10 2>&1 | grep -v amdgpu.ids
done
