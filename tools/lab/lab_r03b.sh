#!/bin/bash
# round 3, GPU batch b: parity of the regenerated predicted-mask fixture, the new kernels' tests, lab timings, bench A/B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "four_iterations or tiny_training" 2>&1 | grep -v Warning | tail -12 > gpurun_out/r03b_small3.log
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention or rowproj or gemm_nt or mlp_fused" 2>&1 | tail -6 > gpurun_out/r03b_kern.log
python tools/rowproj_lab.py > gpurun_out/r03b_rowproj_lab.jsonl 2> gpurun_out/r03b_rowproj_lab.err
for cfg in "1 0" "0 0" "1 1"; do
  set -- $cfg
  CCD_ROWPROJ=$1 CCD_MLP_GELU_POLY=$2 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r03b_bench_rp$1_poly$2.json 2> gpurun_out/r03b_bench_rp$1_poly$2.err
done
cat gpurun_out/r03b_small3.log gpurun_out/r03b_kern.log gpurun_out/r03b_rowproj_lab.jsonl
for f in gpurun_out/r03b_bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], {k:r["by_kind_ms_per_step"][k] for k in ("mlp_fused","gemm_nt_bf16","attention_bwd","gemm_nt_lnbwd","gemm_tn_atomic","gemm_nt_dgelu")})
PY
done
