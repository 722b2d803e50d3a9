#!/bin/bash
# round 5, GPU job 8: A/B of the dropped-MLP-branch cold path in the fused block half (lab_libs/libccd_cold.so) against the shipped library
cd "$(dirname "$0")/../.."
O=gpurun_out
for i in 1 2 3; do
  python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_j8_ship_$i.json
  CCD_HIP_LIB=$PWD/lab_libs/libccd_cold.so python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_j8_cold_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_j8_*.json')):
    d=json.load(open(f)); print(f, d['ms_per_step'], d['config']['final_loss'], d['roofline']['by_kind_ms_per_step'].get('proj_mlp_fused'))
PY
