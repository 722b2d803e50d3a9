#!/bin/bash
# round 6, job 14: fused MLP backward in the step - model A/B test, same-box bench A/B (CCD_FUSE_MLP_BWD = 0 / 1)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "mlp_bwd" 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -8
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); k=d["roofline"]["by_kind_ms_per_step"]; print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], {n: k.get(n) for n in ("proj_mlp_fused", "mlp_bwd_fused", "gemm_nt_dgelu", "gemm_nt_lnbwd", "gemm_tn_pair")})
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(CCD_FUSE_MLP_BWD=0); run r06_j14_two_a
ENVV=(CCD_FUSE_MLP_BWD=1); run r06_j14_fused_a
ENVV=(CCD_FUSE_MLP_BWD=0); run r06_j14_two_b
ENVV=(CCD_FUSE_MLP_BWD=1); run r06_j14_fused_b
