#!/bin/bash
# round 6, job 4: piece table in the kernel arguments (mlp_fused.h) + pipelined head-loss forward: lab timings, tests, step
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python tools/proj_mlp_lab.py 2> gpurun_out/r06_proj_mlp_lab.err | tail -6 > gpurun_out/r06_proj_mlp_lab.jsonl
cat gpurun_out/r06_proj_mlp_lab.jsonl; tail -2 gpurun_out/r06_proj_mlp_lab.err
python tools/head_loss_lab.py 2> /dev/null | tail -1 > gpurun_out/r06_head_loss_lab2.jsonl; cat gpurun_out/r06_head_loss_lab2.jsonl
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "head_loss or mlp_fused or proj_mlp" 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -5
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"), d["roofline"]["by_kind_ms_per_step"].get("proj_mlp_fused"), d["roofline"]["by_kind_ms_per_step"].get("head_loss_fwd"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(X=1); run r06_j4_a
ENVV=(X=1); run r06_j4_b
