#!/bin/bash
# round 5, GPU job 7: tap output + half-height rowproj tiles + fp64-reference gate
cd "$(dirname "$0")/../.."
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "proj_mlp_fused or rowproj or matvec" 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -4
for b in 64 256; do python bench.py --no-cpu-baseline --batch $b 2>/dev/null | tail -1 > $O/r05_j7_b$b.json; done
python -c "
import json
for f in ('$O/r05_j7_b64.json','$O/r05_j7_b256.json'):
    d=json.load(open(f)); k=d['roofline']['by_kind_ms_per_step']; print(f, d['ms_per_step'], d['config']['final_loss'], {n:k[n] for n in ('gemm_nt_bf16','proj_mlp_fused','layernorm_fwd')})"
cat $O/parity_small_step.json
