#!/bin/bash
# round 6, job 16: what pass A of the LayerNorm-backward epilogue costs - lab builds of rowgemm.h without the dbeta column sums (1),
# without both column sums (2), without the x loads (3); results wrong by construction, times only
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
: > gpurun_out/r06_rowgemm_passA_lab.jsonl
for v in base rg1 rg2 rg3 base; do
  if [ $v = base ]; then L=$PWD/ccd_amd/libccd_hip.so; else L=$PWD/ccd_amd/lab_$v.so; fi
  CCD_HIP_LIB=$L RG_QUICK=1 RG_NO_RESID=1 timeout 300 python tools/rowgemm_lab.py --rows 131072 2>/dev/null | grep '"rowgemm": 1, "adma": 1' | grep '"tail": true' | grep -v '"K": 384' | sed "s/^{/{\"lib\": \"$v\", /" | tee -a gpurun_out/r06_rowgemm_passA_lab.jsonl
done
