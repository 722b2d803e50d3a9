#!/bin/bash
# round 6, job 9: a tap's LayerNorm backward folded into the qkv data-gradient product - kernel + model tests, lab timing, step A/B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "lnbwd or fold_tap" 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | grep "passed\|failed\|Error\|error\|assert" | tail -8
python tools/lnbwd_tap_lab.py 2> /dev/null | tee gpurun_out/r06_lnbwd_tap_lab.jsonl
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); k=d["roofline"]["by_kind_ms_per_step"]; print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"), k.get("gemm_nt_lnbwd"), k.get("layernorm_bwd"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(X=1); run r06_j9_fold_a
ENVV=(CCD_FOLD_TAP=0); run r06_j9_sep_a
ENVV=(X=1); run r06_j9_fold_b
ENVV=(CCD_FOLD_TAP=0); run r06_j9_sep_b
