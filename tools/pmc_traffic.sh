#!/bin/bash
# HBM traffic of the step's kernels from PMC counters (MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots"):
# FETCH_SIZE and WRITE_SIZE need separate passes; nothing but --pmc is combined with them.
# usage (GPU box): tools/pmc_traffic.sh <tag>      -> gpurun_out/pmc_<tag>_{fetch,write}/ + gpurun_out/pmc_traffic_<tag>.json
cd /tmp && export TMPDIR=/tmp
tag=${1:-r01}
for c in FETCH_SIZE WRITE_SIZE; do
  d=/root/repo/gpurun_out/pmc_${tag}_$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rocprofv3 --pmc $c --output-format csv -d $d -o b -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $d.log 2>&1
  tail -1 $d.log | cut -c1-200
done
python /root/repo/tools/pmc_traffic.py /root/repo/gpurun_out/pmc_${tag}_fetch/b_counter_collection.csv \
    /root/repo/gpurun_out/pmc_${tag}_write/b_counter_collection.csv > /root/repo/gpurun_out/pmc_traffic_${tag}.json
cat /root/repo/gpurun_out/pmc_traffic_${tag}.json
