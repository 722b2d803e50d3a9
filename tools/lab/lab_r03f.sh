#!/bin/bash
# round 3, GPU batch f: gelu(u) stored by the forward kernel (CCD_STORE_GACT) A/B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mlp_fused or gemm_nt" 2>&1 | tail -4 > gpurun_out/r03f_kern.log
python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "tiny_training or four_iterations" 2>&1 | tail -4 > gpurun_out/r03f_model.log
run() { name=$1; shift; env "$@" > gpurun_out/r03f_bench_$name.json 2> gpurun_out/r03f_bench_$name.err; }
run gact0 CCD_STORE_GACT=0 python bench.py --no-cpu-baseline
run gact1 CCD_STORE_GACT=1 python bench.py --no-cpu-baseline
run gact0_again CCD_STORE_GACT=0 python bench.py --no-cpu-baseline
run gact1_again CCD_STORE_GACT=1 python bench.py --no-cpu-baseline
CCD_STORE_GACT=1 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "tiny_training or four_iterations or micro_batches" 2>&1 | tail -4 > gpurun_out/r03f_model_gact1.log
cat gpurun_out/r03f_kern.log gpurun_out/r03f_model.log gpurun_out/r03f_model_gact1.log
for f in gpurun_out/r03f_bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d.get("roofline",{}).get("by_kind_ms_per_step",{})
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], {x:k.get(x) for x in ("mlp_fused","gemm_nt_dgelu","gemm_nt_lnbwd","gemm_tn_atomic")})
except Exception as e:
    print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
