#!/bin/bash
# round 3, batch m: waits for prefetched rows moved in front of the first store (gemm256.h dgelu epilogue, gemm_row384.h RESID_LN)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_nt or resid_ln or gemm_256 or dgelu or lnbwd" 2>&1 | tail -4
timeout 300 python tools/dgelu_bench.py 2>/dev/null | tee gpurun_out/r03m_dgelu.txt
RG_QUICK=1 timeout 600 python tools/rowgemm_lab.py --rows 131072 2>/dev/null | grep "resid_ln" | tee gpurun_out/r03m_resid.jsonl
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>gpurun_out/r03m_err.log | tail -1 > gpurun_out/r03m_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03m_bench.json"))
r = d["roofline"]["by_kind_ms_per_step"]
print(d["ms_per_step"], d["value"], {k: r[k] for k in ("gemm_nt_dgelu", "gemm_nt_lnbwd", "gemm_nt_resid", "mlp_fused", "attention_bwd", "gemm_nt_bf16") if k in r})
PY
