#!/bin/bash
# round-2 call D: full -m gpu suite, default bench, the N>1 path on one rank, the predicted-mask branch
cd /root/repo; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r02d_gputest.log; cat gpurun_out/r02d_gputest.log | tail -6
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02d_bench.json 2> gpurun_out/r02d_bench.err; tail -c 600 gpurun_out/r02d_bench.json
BENCH_FORCE_DIST=1 MASTER_PORT=29517 timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02d_bench_forcedist.json 2> gpurun_out/r02d_bench_forcedist.err; tail -c 400 gpurun_out/r02d_bench_forcedist.json; tail -3 gpurun_out/r02d_bench_forcedist.err
timeout 300 python bench.py --epoch 30 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02d_bench_epoch30.json 2> gpurun_out/r02d_bench_epoch30.err; tail -c 400 gpurun_out/r02d_bench_epoch30.json
