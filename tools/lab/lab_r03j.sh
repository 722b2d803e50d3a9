#!/bin/bash
# round 3, batch j: where the product phase of rowgemm.h's LayerNorm-backward launch goes - phase stamps (lab build) with the
# activation loads (16), the ring requests (32) or both (48) switched off; then the kernel tests of the release build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CCD_HIP_LIB=$PWD/ccd_amd/libccd_hip_lab.so RG_PHASES=1 RG_LAB=16,32,48 timeout 600 python tools/rowgemm_lab.py --rows 131072 2>/dev/null | grep "^{" > gpurun_out/r03j_rowgemm_phases.jsonl
cat gpurun_out/r03j_rowgemm_phases.jsonl
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "lnbwd or resid_ln" 2>&1 | tail -4
