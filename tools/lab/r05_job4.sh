#!/bin/bash
# round 5, GPU job 4: fused block half, second version (no ring seeks, 72 spills outside the loops)
cd "$(dirname "$0")/../.."
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "proj_mlp_fused" 2>&1 | tail -30 > $O/r05_j4_tests.log
tail -12 $O/r05_j4_tests.log
python tools/proj_mlp_lab.py > $O/r05_j4_lab.jsonl 2>$O/r05_j4_lab.err; cat $O/r05_j4_lab.jsonl
MLP_PHASES=1 CCD_HIP_LIB=$PWD/lab_libs/libccd_mlplab.so python tools/proj_mlp_lab.py 2>/dev/null | grep phases > $O/r05_j4_phases.jsonl; cat $O/r05_j4_phases.jsonl
MLP_PHASES=1 CCD_HIP_LIB=$PWD/lab_libs/libccd_mlplab.so python tools/mlp_lab.py 2>/dev/null | grep phases >> $O/r05_j4_phases.jsonl; tail -2 $O/r05_j4_phases.jsonl
for v in 1 0 1 0; do
  CCD_FUSE_PROJ=$v python bench.py --no-cpu-baseline 2>$O/r05_j4_proj$v.err | tail -1 > $O/r05_j4_proj${v}_$RANDOM.json
done
for f in $O/r05_j4_proj*.json; do echo "$f: $(python -c "
import json
d=json.load(open('$f')); k=d['roofline']['by_kind_ms_per_step']
print(d['ms_per_step'], d['config']['final_loss'], {n:k[n] for n in k if n in ('mlp_fused','proj_mlp_fused','gemm_nt_resid','gemm_nt_bf16','attention_fwd')})")"; done
