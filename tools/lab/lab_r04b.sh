#!/bin/bash
# round 4, call b: rowgemm8 + split-K workspace: parity tests, lab timings, step A/B
mkdir -p gpurun_out/r04b
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "lnbwd or tn_pair" > gpurun_out/r04b/tests.log 2>&1; tail -3 gpurun_out/r04b/tests.log
RG_NO_RESID=1 RG_QUICK=1 python tools/rowgemm_lab.py --rows 131072 > gpurun_out/r04b/rowgemm_lab.jsonl 2>&1; tail -12 gpurun_out/r04b/rowgemm_lab.jsonl
TN3_PAIRS=1 python tools/tn384_lab.py > gpurun_out/r04b/tn_pairs.jsonl 2>&1; cat gpurun_out/r04b/tn_pairs.jsonl
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04b/bench_new.json 2> gpurun_out/r04b/bench_new.err; cut -c1-330 gpurun_out/r04b/bench_new.json
CCD_ROWGEMM8=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04b/bench_rg8off.json 2>/dev/null; cut -c1-330 gpurun_out/r04b/bench_rg8off.json
CCD_TN_WS=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04b/bench_wsoff.json 2>/dev/null; cut -c1-330 gpurun_out/r04b/bench_wsoff.json
