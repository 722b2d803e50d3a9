#!/bin/bash
# round 5, GPU job 6: kernel tests of the new entry points, B = 64 trace, bench
cd "$(dirname "$0")/../.."
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "proj_mlp_fused or matvec" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -4
bash tools/prof_bench.sh r05_b64 --batch 64 > $O/r05_b64_prof.log 2>&1; head -60 $O/r05_b64_steady_state.md
python bench.py --no-cpu-baseline --batch 64 2>/dev/null | tail -1 > $O/r05_j6_b64.json
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_j6_b256.json
python -c "
import json
for f in ('$O/r05_j6_b64.json','$O/r05_j6_b256.json'):
    d=json.load(open(f)); print(f, d['ms_per_step'], d['config']['final_loss'])"
