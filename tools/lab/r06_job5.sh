#!/bin/bash
# round 6, job 5: same-box A/B of the piece table in the kernel arguments (mlp_fused.h) + the sequential head-loss forward again
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in libccd_hip.so lab_mlp_old_prepare.so; do
    echo "== $lib"; CCD_HIP_LIB=$PWD/ccd_amd/$lib python tools/proj_mlp_lab.py 2> /dev/null | tail -4 | head -4
  done
done > gpurun_out/r06_piece_tab_ab.txt
cat gpurun_out/r06_piece_tab_ab.txt
python tools/head_loss_lab.py 2> /dev/null | tail -1 > gpurun_out/r06_head_loss_lab3.jsonl; cat gpurun_out/r06_head_loss_lab3.jsonl
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "head_loss" 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -3
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"), d["roofline"]["by_kind_ms_per_step"].get("proj_mlp_fused"), d["roofline"]["by_kind_ms_per_step"].get("head_loss_fwd"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(X=1); run r06_j5_new_a
ENVV=(CCD_HIP_LIB=$PWD/ccd_amd/lab_mlp_old_prepare.so); run r06_j5_old_a
ENVV=(X=1); run r06_j5_new_b
ENVV=(CCD_HIP_LIB=$PWD/ccd_amd/lab_mlp_old_prepare.so); run r06_j5_old_b
