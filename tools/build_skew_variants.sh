#!/bin/bash
# Development: builds variants of the skewed-tile unit into tools/_probe/libstm_<tag>.so for
# tools/time_libs.sh / tools/ab_sgm.py.  Usage: tools/build_skew_variants.sh tag1 "-DFLAG ..." tag2 "..." ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/stereo-matching_amd/csrc
make -C $C -j8 2>&1 | grep -E "error" || true
mkdir -p $ROOT/tools/_probe /tmp/isa
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I$ROOT/include -I$C"
OTHERS=$(ls $C/*.o | grep -v "stm_sgm_skew_d")
cp $ROOT/stereo-matching_amd/stereomatch/libstereomatch_hip.so $ROOT/tools/_probe/libstm_prod.so
while [ $# -ge 2 ]; do
  TAG=$1; DEFS=$2; shift 2
  ( /opt/rocm/bin/hipcc $FL $DEFS -c $C/stm_sgm_skew_d128.hip -o /tmp/isa/skew128_$TAG.o 2>&1 | grep -E "error" || true
    /opt/rocm/bin/hipcc $FL $DEFS -c $C/stm_sgm_skew_d256.hip -o /tmp/isa/skew256_$TAG.o 2>&1 | grep -E "error" || true
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_probe/libstm_$TAG.so $OTHERS /tmp/isa/skew128_$TAG.o /tmp/isa/skew256_$TAG.o ) &
done
wait
ls -la $ROOT/tools/_probe/*.so | awk '{print $5,$9}'
