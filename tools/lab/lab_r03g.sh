#!/bin/bash
# round 3, GPU batch g: unequal contraction slices in the weight-gradient kernel (gemm_tn384_skew)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm_tn" 2>&1 | tail -3 > gpurun_out/r03g_kern.log
python - > gpurun_out/r03g_tn_lab.jsonl 2> gpurun_out/r03g_tn_lab.err <<'PY'
import json, sys, torch
sys.path.insert(0, ".")
from ccd_amd import ops
from tools.mlp_lab import timeit
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0); BF = torch.bfloat16
R = 131072
gb = torch.randn(R, 384, generator=g).to(BF).to(dev); gact = torch.randn(R, 1536, generator=g).to(BF).to(dev)
du = torch.randn(R, 1536, generator=g).to(BF).to(dev); y2 = torch.randn(R, 384, generator=g).to(BF).to(dev)
att = torch.randn(R, 384, generator=g).to(BF).to(dev); dqkv = torch.randn(R, 1152, generator=g).to(BF).to(dev)
c1 = torch.zeros(384, 1536, device=dev); c2 = torch.zeros(1536, 384, device=dev); c3 = torch.zeros(384, 384, device=dev); c4 = torch.zeros(1152, 384, device=dev)
for pct in (0, 8, 14, 20, 28, 36):
    with ops.policy(gemm_tn384_skew=pct):
        a = timeit(lambda: ops.gemm_tn_pair(gb, gact, c1, du, y2, c2))
        b = timeit(lambda: ops.gemm_tn_pair(gb, att, c3, dqkv, y2, c4))
    print(json.dumps({"gemm_tn384_skew_pct": pct, "mlp_pair_ms": round(a, 4), "attention_pair_ms": round(b, 4)}), flush=True)
PY
run() { name=$1; shift; env "$@" > gpurun_out/r03g_bench_$name.json 2> gpurun_out/r03g_bench_$name.err; }
run skew0 CCD_GEMM_TN384_SKEW=0 python bench.py --no-cpu-baseline
run skew14 CCD_GEMM_TN384_SKEW=14 python bench.py --no-cpu-baseline
run skew22 CCD_GEMM_TN384_SKEW=22 python bench.py --no-cpu-baseline
run skew0_again CCD_GEMM_TN384_SKEW=0 python bench.py --no-cpu-baseline
cat gpurun_out/r03g_kern.log gpurun_out/r03g_tn_lab.jsonl; tail -3 gpurun_out/r03g_tn_lab.err
for f in gpurun_out/r03g_bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d.get("roofline",{}).get("by_kind_ms_per_step",{})
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], {x:k.get(x) for x in ("gemm_tn_atomic","mlp_fused","gemm_nt_dgelu")})
except Exception as e:
    print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
