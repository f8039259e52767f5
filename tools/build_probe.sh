#!/bin/bash
# Builds the product library and the diagnostic (stamped) variant tools/_probe/libstm_stamps.so;
# options: usage (register / scratch use), recompute, ablate (timing probes, see below),
# microbench (tools/valu_microbench.hip, tools/h2_dpp_check.hip).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/stereo-matching_amd/csrc
make -C $C -j8 2>&1 | grep -E "error|warning:" || true
mkdir -p $ROOT/tools/_probe /tmp/isa
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I$ROOT/include -I$C"
/opt/rocm/bin/hipcc $FL -DSTM_TILE_STAMPS -c $C/stm_sgm_tiled.hip -o /tmp/isa/tiled_stamps.o 2>&1 | grep -E "error|warning:" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_probe/libstm_stamps.so $C/stm_api.o $C/stm_wta.o $C/stm_ssd.o $C/stm_birchfield.o $C/stm_sgm.o /tmp/isa/tiled_stamps.o $C/stm_dp.o $C/stm_sgm_exact.o
if [ "$1" = "recompute" ]; then
  # the horizontal walk with the arithmetic of an in-walk SSD recomputation (timing probe)
  /opt/rocm/bin/hipcc $FL -DSTM_HWALK_RECOMPUTE_PROBE -c $C/stm_sgm_tiled.hip -o /tmp/isa/tiled_recompute.o 2>&1 | grep -E "error|warning:" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_probe/libstm_recompute.so $C/stm_api.o $C/stm_wta.o $C/stm_ssd.o $C/stm_birchfield.o $C/stm_sgm.o /tmp/isa/tiled_recompute.o $C/stm_dp.o $C/stm_sgm_exact.o
fi
if [ "$1" = "ablate" ]; then
  # the tile kernel without its sum updates (1), its chains (2), both (3), its final output and
  # winner-takes-all (4): timing ablations, results are wrong (tools/time_libs.sh "ablate1 ...")
  for A in 1 2 3 4; do
    /opt/rocm/bin/hipcc $FL -DSTM_TILE_ABLATE=$A -c $C/stm_sgm_tiled.hip -o /tmp/isa/tiled_ablate$A.o 2>&1 | grep -E "error|warning:" || true
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_probe/libstm_ablate$A.so $C/stm_api.o $C/stm_wta.o $C/stm_ssd.o $C/stm_birchfield.o $C/stm_sgm.o /tmp/isa/tiled_ablate$A.o $C/stm_dp.o $C/stm_sgm_exact.o
  done
fi
if [ "$1" = "microbench" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-result -o $ROOT/tools/_probe/valu_microbench $ROOT/tools/valu_microbench.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-result -o $ROOT/tools/_probe/h2_dpp_check $ROOT/tools/h2_dpp_check.hip
fi
if [ "$1" = "usage" ]; then
  /opt/rocm/bin/hipcc $FL --cuda-device-only -S $C/stm_sgm_tiled.hip -o /tmp/isa/tiled.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|Spill|Scratch" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - - | sed 's/_ZN3stm12_GLOBAL__N_1//; s/EvPKfS3.*E\t/\t/' | grep -E "ILi[24]E"
fi
cp $ROOT/stereo-matching_amd/stereomatch/libstereomatch_hip.so $ROOT/tools/_probe/libstm_prod.so  # for tools/time_libs.sh
ls -la $ROOT/stereo-matching_amd/stereomatch/libstereomatch_hip.so $ROOT/tools/_probe/libstm_stamps.so | awk '{print $6,$7,$8,$9}'
