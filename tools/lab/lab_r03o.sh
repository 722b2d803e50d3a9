#!/bin/bash
# round 3, batch o: mlp_fused with the tile's DropPath scale through the scalar cache - kernel tests, lab timing, the step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "mlp_fused" 2>&1 | tail -3
timeout 300 python tools/mlp_lab.py 2>/dev/null | grep "^{" | head -6 | tee gpurun_out/r03o_mlp_lab.jsonl
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>gpurun_out/r03o_err.log | tail -1 > gpurun_out/r03o_bench_$i.json
python - <<PY
import json
d = json.load(open("gpurun_out/r03o_bench_$i.json"))
r = d["roofline"]["by_kind_ms_per_step"]
print(d["ms_per_step"], d["value"], {k: r[k] for k in ("gemm_nt_dgelu", "gemm_nt_lnbwd", "gemm_nt_resid", "mlp_fused", "attention_bwd", "gemm_nt_bf16", "gemm_tn_atomic") if k in r})
PY
done
