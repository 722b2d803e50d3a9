#!/bin/bash
# round 6, job 13: fused MLP backward (mlp_bwd.h) - kernel tests, lab timing against the two launches
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "mlp_bwd" 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -15
timeout 300 python tools/mlp_bwd_lab.py 2> gpurun_out/r06_mlp_bwd_lab.err | tee gpurun_out/r06_mlp_bwd_lab.jsonl
tail -5 gpurun_out/r06_mlp_bwd_lab.err
