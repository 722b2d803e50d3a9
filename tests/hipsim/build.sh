#!/bin/bash
# Build tests/hipsim/libccd_sim.so: same kernels + ABI, executed by the fiber-based CPU SIMT executor.
set -euo pipefail
cd "$(dirname "$0")"
CXX=${SIM_CXX:-/opt/rocm/lib/llvm/bin/clang++}
$CXX -O2 -std=c++17 -fPIC -shared -pthread -ffp-contract=off -Wno-unused-value -Wno-unknown-attributes -Wno-psabi \
    ccd_sim.cpp hipsim.cpp -o libccd_sim.so
echo "built $(realpath libccd_sim.so)"
