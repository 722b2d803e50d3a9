#!/bin/bash
# rocprofv3 kernel trace of bench.py (GPU box) -> steady-state summary + the whole-run stats csv under gpurun_out/
# usage: tools/prof_bench.sh <tag> [bench.py arguments]
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
d=/root/repo/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o bench -- python /root/repo/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $d.log 2>&1
tail -1 $d.log | cut -c1-300
python /root/repo/tools/prof_summary.py $d/bench_kernel_trace.csv --steps 2 --sequence /root/repo/gpurun_out/${tag}_sequence.md > /root/repo/gpurun_out/${tag}_steady_state.md
cp $d/bench_kernel_stats.csv /root/repo/gpurun_out/${tag}_rocprofv3_kernel_stats.csv 2>/dev/null
head -45 /root/repo/gpurun_out/${tag}_steady_state.md
rm -rf $d
