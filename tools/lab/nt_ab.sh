#!/bin/bash
# A/B of the non-temporal stream table (prelude_hip.h: CCD_NT): one bench.py run per library built with -DCCD_NT=<mask> under lab_libs/.
# usage (GPU box): tools/lab/nt_ab.sh <mask> [<mask> ...]   -> one JSON line per mask on stdout
cd "$(dirname "$0")/../.."
for m in "$@"; do
  CCD_HIP_LIB=$PWD/lab_libs/libccd_nt_$m.so python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['by_kind_ms_per_step']
print(json.dumps({'CCD_NT':'$m','ms_per_step':d['ms_per_step'],**{n:k[n] for n in ('mlp_fused','gemm_nt_lnbwd','gemm_nt_dgelu','gemm_nt_bf16','gemm_tn_atomic','gemm_nt_resid','attention_fwd','attention_bwd')}}))"
done
