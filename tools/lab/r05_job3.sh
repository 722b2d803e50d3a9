#!/bin/bash
# round 5, GPU job 3: the fused block half (proj + LN2 + MLP): GPU parity tests, then the step with and without it
cd "$(dirname "$0")/../.."
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "proj_mlp_fused or test_mlp_fused" 2>&1 | tail -5 > $O/r05_j3_tests.log
cat $O/r05_j3_tests.log
for v in 1 0 1 0; do
  CCD_FUSE_PROJ=$v python bench.py --no-cpu-baseline 2>$O/r05_j3_proj$v.err | tail -1 > $O/r05_j3_proj${v}_$RANDOM.json
done
for f in $O/r05_j3_proj*.json; do echo "$f: $(python -c "
import json
d=json.load(open('$f')); k=d['roofline']['by_kind_ms_per_step']
print(d['ms_per_step'], d['config']['final_loss'], {n:k[n] for n in k if n in ('mlp_fused','proj_mlp_fused','gemm_nt_resid','gemm_nt_bf16','attention_fwd')})")"; done
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "tiny or small_step or smoke" 2>&1 | tail -5
