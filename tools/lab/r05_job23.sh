#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "small_ops or head_pieces or seg_loss or optimizer or small_step or tiny or checkpoint" > gpurun_out/r05_job23_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r05_job23_tests.log | tail -2
bash tools/prof_bench.sh r05_micro > gpurun_out/r05_micro.out 2>&1
grep -n "weightnorm_fwd\|mirror_bf16\|seg_loss\|^wall" gpurun_out/r05_micro_steady_state.md | cut -c1-160
