#!/bin/bash
# round 6, job 6: head + loss with two workgroups per CU
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python tools/head_loss_lab.py 2> /dev/null | tail -1 > gpurun_out/r06_head_loss_lab4.jsonl; cat gpurun_out/r06_head_loss_lab4.jsonl
python tools/head_loss_lab.py --m 412 --max-rows 3328 2> /dev/null | tail -1 >> gpurun_out/r06_head_loss_lab4.jsonl; tail -1 gpurun_out/r06_head_loss_lab4.jsonl
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "head_loss" 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -3
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"), d["roofline"]["by_kind_ms_per_step"].get("head_loss_fwd"), d["roofline"]["by_kind_ms_per_step"].get("head_loss_bwd"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(X=1); run r06_j6_a
ENVV=(X=1); run r06_j6_b
