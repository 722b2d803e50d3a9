#!/bin/bash
# round 3, batch i: rowgemm.h with hand-tracked activation loads (no compiler vmcnt(0) in the main loop) and a spill-free
# RESID_LN epilogue - kernel tests, lab timing, in-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "lnbwd or resid_ln or mlp_fused or rowproj" 2>&1 | tail -6 > gpurun_out/r03i_tests.log
cat gpurun_out/r03i_tests.log
RG_QUICK=1 timeout 600 python tools/rowgemm_lab.py --rows 131072 2>/dev/null | grep "^{" > gpurun_out/r03i_rowgemm_lab.jsonl
cat gpurun_out/r03i_rowgemm_lab.jsonl
: > gpurun_out/r03i_bench_ab.jsonl
for RG in 1 2; do
  CCD_ROWGEMM=$RG timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>gpurun_out/r03i_err_$RG.log | tail -1 >> gpurun_out/r03i_bench_ab.jsonl
  tail -2 gpurun_out/r03i_err_$RG.log
done
python - <<'PY'
import json
for l in open("gpurun_out/r03i_bench_ab.jsonl"):
    try: d = json.loads(l)
    except Exception: print("bad line", l[:200]); continue
    r = d["roofline"]["by_kind_ms_per_step"]
    print(d["ms_per_step"], d["value"], {k: r[k] for k in ("gemm_nt_lnbwd", "gemm_nt_resid", "mlp_fused") if k in r})
PY
