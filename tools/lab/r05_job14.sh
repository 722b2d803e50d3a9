#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "cls_tail or seghead" > gpurun_out/r05_job14_tests.log 2>&1
tail -3 gpurun_out/r05_job14_tests.log
for lab in 0 1 2 3 4; do timeout 300 python tools/cls_tail_lab.py --fused-only --lab $lab 2>&1 | tail -1 | tee -a gpurun_out/r05_cls_tail_lab.jsonl; done
