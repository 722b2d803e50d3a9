#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_datapipe_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "augment or datapipe or lmdb or finetune_dataset or multi_launch" > gpurun_out/r05_job22_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r05_job22_tests.log | tail -2
