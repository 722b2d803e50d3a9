#!/bin/bash
# round 4, call c: LayerNorm-backward product with the activation rows by LDS-DMA (rowgemm.h ADMA): parity, lab timing, step A/B
mkdir -p gpurun_out/r04c
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "lnbwd or resid_ln or tn_pair" > gpurun_out/r04c/tests.log 2>&1; tail -3 gpurun_out/r04c/tests.log
RG_QUICK=1 python tools/rowgemm_lab.py --rows 131072 > gpurun_out/r04c/rowgemm_lab.jsonl 2>&1; grep -v "rowgemm\": 0" gpurun_out/r04c/rowgemm_lab.jsonl | tail -12
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04c/bench_new.json 2> gpurun_out/r04c/bench_new.err; cut -c1-330 gpurun_out/r04c/bench_new.json
CCD_ROWGEMM_ADMA=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04c/bench_admaoff.json 2>/dev/null; cut -c1-330 gpurun_out/r04c/bench_admaoff.json
