#!/bin/bash
# On the GPU box: kernel times (rocprofv3 --kernel-trace --stats) of the fused aggregation of
# probe builds of the library.  tools/time_libs.sh "<lib tags>" "<H W D B>" ["<H W D B>" ...]
# Appends to gpurun_out/time_libs.log as it goes.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LOG=$ROOT/gpurun_out/time_libs.log
mkdir -p $ROOT/gpurun_out
TAGS=$1; shift
cd /tmp && export TMPDIR=/tmp
for SHAPE in "$@"; do
  for L in $TAGS; do
    D=/tmp/tl_${L}_$(echo $SHAPE | tr ' ' _)
    rm -rf $D
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python3 $ROOT/tools/run_lib.py $ROOT/tools/_probe/libstm_$L.so $SHAPE > $D.log 2>&1 || { echo "== $L $SHAPE FAILED" >> $LOG; cp $D.log $ROOT/gpurun_out/time_libs_fail_${L}.log; tail -5 $D.log >> $LOG; exit 1; }
    echo "== $L $SHAPE" >> $LOG
    python3 - $D >> $LOG <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sgm_" in r["Name"]:
            n = r["Name"].replace("void stm::(anonymous namespace)::", "").split("(")[0]
            print("  %-40s calls %4s avg_us %10.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
tail -60 $LOG
