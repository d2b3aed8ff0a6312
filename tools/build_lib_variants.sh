#!/bin/bash
# Variant builds of libabcdez_hip.so for same-box A/B runs (ABCDEZ_HIP_LIB=<path> python bench.py ...):
#   VARIANTS="name|make EXTRA flags" ... -> abcdez.jl_amd/lib/variants/libabcdez_hip_<name>.so
set -e
cd "$(dirname "$0")/../abcdez.jl_amd/csrc"
VARIANTS=${VARIANTS:-"
1p|-DABZ_SWEEP_ONE_PHASE
2p_w4|-DABZ_SWEEP_WAVES=4
2p_w3|-DABZ_SWEEP_WAVES=3
"}
while IFS='|' read -r name flags; do
  [ -z "$name" ] && continue
  make -s -j6 BUILD=build_$name OUT=../lib/variants/libabcdez_hip_$name.so EXTRA="$flags"
  echo "built lib/variants/libabcdez_hip_$name.so ($flags)"
done <<< "$VARIANTS"
