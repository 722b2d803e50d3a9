#!/bin/bash
# round 6, job 12: packed fp32 math in the attention kernels' softmax - tests, kernel times in the step
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | grep "passed\|failed\|Error\|error\|assert" | tail -8
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); k=d["roofline"]["by_kind_ms_per_step"]; print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"), k.get("attention_fwd"), k.get("attention_bwd"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(X=1); run r06_j12_new_a
ENVV=(CCD_HIP_LIB=$PWD/ccd_amd/lab_old_attention.so); run r06_j12_old_a
ENVV=(X=1); run r06_j12_new_b
ENVV=(CCD_HIP_LIB=$PWD/ccd_amd/lab_old_attention.so); run r06_j12_old_b
