#!/bin/bash
# round 6, job 17: patch embedding with scalar pixel loads + fused multiply-adds, positional resample with 32 loads in flight -
# kernel tests, same-box bench A/B against the previous library
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "patch_embed or small_ops" 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -5
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); k=d["roofline"]["by_kind_ms_per_step"]; print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], {n: v for n, v in k.items() if "patch" in n or "small" in n or "pos" in n})
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(X=1); run r06_j17_new_a
ENVV=(CCD_HIP_LIB=$PWD/ccd_amd/lab_prev.so); run r06_j17_prev_a
ENVV=(X=1); run r06_j17_new_b
ENVV=(CCD_HIP_LIB=$PWD/ccd_amd/lab_prev.so); run r06_j17_prev_b
ENVV=(X=1); run r06_j17_new_c
ENVV=(CCD_HIP_LIB=$PWD/ccd_amd/lab_prev.so); run r06_j17_prev_c
