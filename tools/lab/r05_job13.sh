#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "cls_tail or seghead" > gpurun_out/r05_job13_tests.log 2>&1
tail -3 gpurun_out/r05_job13_tests.log
timeout 600 python tools/cls_tail_lab.py 2>&1 | tail -2 | tee -a gpurun_out/r05_cls_tail_lab.jsonl
