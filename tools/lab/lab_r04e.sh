#!/bin/bash
# round 4, call e: where does the N > 1 code path cost time on one rank?  kernel traces of the plain and the forced-distributed step
mkdir -p gpurun_out/r04e
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04e/bench_plain.json 2>/dev/null; cut -c1-200 gpurun_out/r04e/bench_plain.json
BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04e/bench_forcedist.json 2>/dev/null; cut -c1-200 gpurun_out/r04e/bench_forcedist.json
tools/prof_bench.sh r04e_plain > gpurun_out/r04e/prof_plain.log 2>&1
BENCH_FORCE_DIST=1 tools/prof_bench.sh r04e_forcedist > gpurun_out/r04e/prof_forcedist.log 2>&1
head -30 gpurun_out/r04e_forcedist_steady_state.md
