cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/time_libs.log
bash tools/time_libs.sh "B" "375 450 128 256" "1080 1920 256 8" > /dev/null 2>&1
cat gpurun_out/time_libs.log
python tools/sgm_forms.py 375 450 128 256
