#!/bin/bash
# round 6, job 11: the head's convolutions on the 256-row LDS-DMA tile (gemm256.h CONV) - tests, step A/B against the 128-square kernel
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv or seghead or cls_tail" 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | grep "passed\|failed\|Error\|error\|assert" | tail -8
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); k=d["roofline"]["by_kind_ms_per_step"]; print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"), k.get("conv_gemm"), k.get("conv_wgrad"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(X=1); run r06_j11_c256_a
ENVV=(CCD_CONV_256=0); run r06_j11_c128_a
ENVV=(CCD_CONV_256=2); run r06_j11_c256w_a
ENVV=(X=1); run r06_j11_c256_b
ENVV=(CCD_CONV_256=0); run r06_j11_c128_b
ENVV=(CCD_CONV_256=2); run r06_j11_c256w_b
