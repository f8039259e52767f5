cd /tmp && export TMPDIR=/tmp
for B in 1 8; do
rm -rf /tmp/dpk$B; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dpk$B -o r -- python3 $GRAFT_REPO_ROOT/tools/_probe/dp8.py $B > /tmp/dpk$B.log 2>&1
echo "== B=$B"; python3 - /tmp/dpk$B <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dp_" in r["Name"]:
            print("  %-60s calls %4s avg_us %10.1f" % (r["Name"].replace("void stm::(anonymous namespace)::", "").split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
