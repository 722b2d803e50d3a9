#!/bin/bash
# round 6, job 15: 16 bytes of scratch in rowproj / attention_fwd (scratch classes) - same-box bench A/B against the previous library
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); k=d["roofline"]["by_kind_ms_per_step"]; print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], {n: k.get(n) for n in ("proj_mlp_fused", "gemm_nt_bf16", "attention_fwd")})
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(X=1); run r06_j15_pad_a
ENVV=(CCD_HIP_LIB=$PWD/ccd_amd/lab_prev.so); run r06_j15_prev_a
ENVV=(X=1); run r06_j15_pad_b
ENVV=(CCD_HIP_LIB=$PWD/ccd_amd/lab_prev.so); run r06_j15_prev_b
ENVV=(X=1); run r06_j15_pad_c
ENVV=(CCD_HIP_LIB=$PWD/ccd_amd/lab_prev.so); run r06_j15_prev_c
