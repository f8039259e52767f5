#!/bin/bash
# Runs on the GPU box: kernel trace + stats of bench.py, then FETCH_SIZE and WRITE_SIZE in
# separate counter passes (MI355X guide: they do not fit one pass), into gpurun_out/<tag>_*.
# Afterwards, in the build container:
#   STM_PROF_PREFIX=<tag> python profiles/summarize.py <tag> gpurun_out/<tag>_kt gpurun_out/<tag>_fetch gpurun_out/<tag>_write <teddy pairs> <1080p pairs>
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
ARGS="--steps 3 --warmup 1 --also-steps 2 --no-cpu-baseline --no-config3 --no-latency"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o $TAG -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -o $TAG -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -o $TAG -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/${TAG}_write.err
ls $OUT/${TAG}_kt $OUT/${TAG}_fetch $OUT/${TAG}_write
