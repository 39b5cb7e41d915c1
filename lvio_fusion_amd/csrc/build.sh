#!/bin/bash
# Builds lvio_fusion_amd/liblvf_hip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../liblvf_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRCS=$(ls "$HERE"/*.hip)
# bounded: an optimiser pathology (see the asm barrier in k_chol_factor_panel) must fail the build, not hang it
timeout "${LVF_BUILD_TIMEOUT:-1800}" "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-fast-math -Wall -Wno-unused-function \
  -I"$HERE/../../include" $SRCS -o "$OUT" "$@"
echo "built $OUT"
