#!/bin/bash
# round 5, job 10: Snowflakes / Rain + the layer farm on the GPU box: datapipe tests, the augment kernel's parity rows, the view maker's steady state
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_datapipe_gpu.py tests/test_kernels_gpu.py -m gpu -q -k "augment or pipeline or data_aug" > gpurun_out/r05_job10_tests.log 2>&1
tail -5 gpurun_out/r05_job10_tests.log
timeout 600 python tools/viewmaker_bench.py > gpurun_out/r05_viewmaker.jsonl 2> gpurun_out/r05_viewmaker.err
cat gpurun_out/r05_viewmaker.jsonl; tail -3 gpurun_out/r05_viewmaker.err
timeout 600 python tools/viewmaker_bench.py --workers 1 --iters 4 >> gpurun_out/r05_viewmaker.jsonl 2>> gpurun_out/r05_viewmaker.err
tail -2 gpurun_out/r05_viewmaker.jsonl
