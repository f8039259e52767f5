for n in 1 2 3; do
  export STM_TILE_WGS_PER_CU=$n
  echo "wgs per cu $n" >> $GRAFT_REPO_ROOT/gpurun_out/time_libs.log
  tools/time_libs.sh "prod" "375 450 128 256" > /dev/null
done
tail -20 $GRAFT_REPO_ROOT/gpurun_out/time_libs.log
