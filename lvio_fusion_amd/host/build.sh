#!/bin/bash
# Builds the C++ host-interface self-test against the C-ABI library (plain g++: the adapter has no HIP in its headers).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$HERE/../.."
CXX="${CXX:-g++}"
"$CXX" -O2 -std=c++17 -Wall -Wextra -Wno-unused-parameter -I"$ROOT/include" -I"$HERE" "$HERE/adapter_selftest.cpp" \
  -L"$HERE/.." -llvf_hip -Wl,-rpath,'$ORIGIN/..' -o "$HERE/adapter_selftest"
echo "built $HERE/adapter_selftest"
# the C++ multi-GPU driver of the candidate loop (one thread per device inside a process, one RCCL all-gather across processes)
"$CXX" -O2 -std=c++17 -Wall -Wextra -pthread -I"$ROOT/include" "$HERE/relocalize_driver.cpp" \
  -L"$HERE/.." -llvf_hip -Wl,-rpath,'$ORIGIN/..' -o "$HERE/relocalize_driver"
echo "built $HERE/relocalize_driver"
