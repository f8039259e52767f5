#!/bin/bash
# Builds the product library and the diagnostic (stamped) variant tools/_probe/libstm_stamps.so;
# options: usage (register / scratch use of the tiled kernels), microbench
# (tools/valu_microbench.hip, tools/h2_dpp_check.hip).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/stereo-matching_amd/csrc
make -C $C -j8 2>&1 | grep -E "error|warning:" || true
mkdir -p $ROOT/tools/_probe /tmp/isa
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I$ROOT/include -I$C"
TILED="stm_sgm_tiled stm_sgm_sweep stm_sgm_hwalk stm_sgm_bwalk stm_sgm_tile stm_sgm_tile_h2"
if [ "$1" != "usage" ]; then
  OBJS=""
  for f in $TILED; do
    /opt/rocm/bin/hipcc $FL -DSTM_TILE_STAMPS -c $C/$f.hip -o /tmp/isa/${f}_stamps.o 2>&1 | grep -E "error|warning:" || true &
    OBJS="$OBJS /tmp/isa/${f}_stamps.o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_probe/libstm_stamps.so $C/stm_api.o $C/stm_wta.o $C/stm_ssd.o $C/stm_birchfield.o $C/stm_sgm.o $OBJS $C/stm_dp.o $C/stm_sgm_exact.o
fi
if [ "$1" = "microbench" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-result -o $ROOT/tools/_probe/valu_microbench $ROOT/tools/valu_microbench.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-result -o $ROOT/tools/_probe/h2_dpp_check $ROOT/tools/h2_dpp_check.hip
fi
if [ "$1" = "usage" ]; then
  for f in stm_sgm_sweep stm_sgm_hwalk stm_sgm_tile stm_sgm_tile_h2 ${2}; do
    /opt/rocm/bin/hipcc $FL --cuda-device-only -S $C/$f.hip -o /tmp/isa/$f.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|Spill|Scratch" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - - | sed 's/_ZN3stm12_GLOBAL__N_1//; s/EvPKfS3.*E\t/\t/' &
  done
  wait
fi
cp $ROOT/stereo-matching_amd/stereomatch/libstereomatch_hip.so $ROOT/tools/_probe/libstm_prod.so  # for tools/time_libs.sh
ls -la $ROOT/stereo-matching_amd/stereomatch/libstereomatch_hip.so $ROOT/tools/_probe/libstm_stamps.so | awk '{print $6,$7,$8,$9}'
