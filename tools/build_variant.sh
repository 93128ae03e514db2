#!/bin/bash
# Build a variant of the engine library with extra compiler flags (timing-only experiments, A/B candidates):
#   tools/build_variant.sh NAME [-DFLAG ...]   ->  chimeralm_amd/csrc/libclm_NAME.so   (run with CLM_LIB=<that path>, tools/ab.sh)
# Timing-only switches live in chimeralm_amd/csrc/clm_lab.h and need -DCLM_LAB next to the switch: e.g. `-DCLM_LAB -DCLM_EXP_NOLN`.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/chimeralm_amd/csrc; name=$1; shift
T=$(mktemp -d)
srcs="clm_api gemm gemm16 tail32 hyena_conv head pad_prefix lone_token attention tf_model tf_fp32"
for s in $srcs; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-function -Wno-pass-failed "$@" -I$R/include -I$C -c $C/$s.hip -o $T/$s.o &
done
for s in bam_feeder bam_filter; do
  hipcc -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-function "$@" -I$R/include -I$C -c $C/$s.cpp -o $T/$s.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libclm_$name.so $T/*.o -lz -lpthread
rm -rf $T
echo $C/libclm_$name.so
