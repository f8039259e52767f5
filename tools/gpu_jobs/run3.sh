set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/time_libs.log
export STM_SWEEP_PLAIN_LAUNCH=1
bash tools/time_libs.sh "A B" "375 450 128 256" "1080 1920 256 8" > /dev/null 2>&1
unset STM_SWEEP_PLAIN_LAUNCH
cp gpurun_out/time_libs.log gpurun_out/r3c_time_plain.log
bash tools/time_libs.sh "B" "375 450 128 64" > /dev/null 2>&1
cat gpurun_out/time_libs.log
