#!/bin/bash
# round 6, job 2: the bf16 residual-gradient stream - kernel / model tests, step A/B, the parity suite with the stream switched on
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "layernorm or gemm_lnbwd or patch_embed or g_bf16" 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -12 > gpurun_out/r06_job2_tests.log
cat gpurun_out/r06_job2_tests.log
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(CCD_G_BF16=1); run r06_j2_g16_a
ENVV=(CCD_G_BF16=0); run r06_j2_g32_a
ENVV=(CCD_G_BF16=1); run r06_j2_g16_b
ENVV=(CCD_G_BF16=0); run r06_j2_g32_b
# the model-level parity suite with the bf16 stream on (reports land in gpurun_out/parity_*.json)
CCD_G_BF16=1 python -m pytest tests/test_model_gpu.py -m gpu -q 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -25 > gpurun_out/r06_job2_parity_g16.log
cat gpurun_out/r06_job2_parity_g16.log
mkdir -p gpurun_out/g16 && cp gpurun_out/parity_*.json gpurun_out/g_bf16_vs_fp32.json gpurun_out/g16/ 2>/dev/null
