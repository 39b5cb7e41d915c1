#!/bin/bash
# Builds the C++ host-interface self-test against the C-ABI library (plain g++: the adapter has no HIP in its headers).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$HERE/../.."
CXX="${CXX:-g++}"
"$CXX" -O2 -std=c++17 -Wall -Wextra -Wno-unused-parameter -I"$ROOT/include" -I"$HERE" "$HERE/adapter_selftest.cpp" \
  -L"$HERE/.." -llvf_hip -Wl,-rpath,'$ORIGIN/..' -o "$HERE/adapter_selftest"
echo "built $HERE/adapter_selftest"
