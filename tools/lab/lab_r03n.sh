#!/bin/bash
# round 3, batch n: A/B of `needed_here` (waits in front of the first store) - release library vs a lab build without it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
for L in ccd_amd/libccd_hip.so ccd_amd/libccd_hip_lab.so; do
  echo "== $L"
  CCD_HIP_LIB=$PWD/$L timeout 300 python tools/dgelu_bench.py 2>/dev/null
  CCD_HIP_LIB=$PWD/$L RG_QUICK=1 timeout 600 python tools/rowgemm_lab.py --rows 131072 2>/dev/null | grep "resid_ln"
done
done 2>&1 | tee gpurun_out/r03n_needed_here_ab.txt
