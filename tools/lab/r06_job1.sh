#!/bin/bash
# round 6, job 1: the fused head + loss kernels on the GPU - kernel / model tests, lab timings, step A/B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "head_loss or dino_loss or no_grad_train or smoke" 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -12 > gpurun_out/r06_job1_tests.log
cat gpurun_out/r06_job1_tests.log
python tools/head_loss_lab.py 2> gpurun_out/r06_head_loss_lab.err | tail -1 > gpurun_out/r06_head_loss_lab.jsonl
python tools/head_loss_lab.py --m 412 --max-rows 3328 2>> gpurun_out/r06_head_loss_lab.err | tail -1 >> gpurun_out/r06_head_loss_lab.jsonl
cat gpurun_out/r06_head_loss_lab.jsonl; tail -3 gpurun_out/r06_head_loss_lab.err
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(X=1); run r06_j1_fused_a
ENVV=(CCD_FUSE_HEAD_LOSS=0); run r06_j1_unfused_a
ENVV=(X=1); run r06_j1_fused_b
ENVV=(CCD_FUSE_HEAD_LOSS=0); run r06_j1_unfused_b
ENVV=(X=1); run r06_j1_fused_b64 --batch 64
ENVV=(CCD_FUSE_HEAD_LOSS=0); run r06_j1_unfused_b64 --batch 64
