#!/bin/bash
# round 5, GPU job 9: the data-pipeline tests on the device (new augmenter members) before the evidence re-run
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_datapipe_gpu.py -x -q -k "augment or datapipe or lmdb or views or kmeans" 2>&1 | tail -12
