#!/bin/bash
# Build libccd_hip.so for gfx950 in-tree (ccd_amd/libccd_hip.so).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${CCD_OUT:-../libccd_hip.so}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
    -Wno-unused-value ${CCD_EXTRA_FLAGS:-} ccd_hip.hip -o $OUT
echo "built $(realpath $OUT)"
