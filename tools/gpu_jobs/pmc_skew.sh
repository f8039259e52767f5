# PMC counters of the skewed-tile form's two kernels (64 or $1 Teddy pairs), one rocprofv3 pass per set.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-64}
SHAPE=${2:-"375 450 128"}
LOG=$R/gpurun_out/pmc_skew.log
mkdir -p $R/gpurun_out; rm -f $LOG
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  D=/tmp/pmc_$(echo $SET | cut -c1-12 | tr ' ' _); rm -rf $D
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -o r -- python3 $R/tools/run_lib.py $R/stereo-matching_amd/stereomatch/libstereomatch_hip.so $SHAPE $B skew > $D.log 2>&1 || { tail -5 $D.log; }
  python3 - $D <<'PY' >> $LOG
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void stm::(anonymous namespace)::", "").replace("stm::(anonymous namespace)::", "").split("(")[0]
        if "sgm_" in n: agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, cs in agg.items():
        print(n, {k: round(sum(v) / len(v)) for k, v in cs.items()})
PY
done
cat $LOG
