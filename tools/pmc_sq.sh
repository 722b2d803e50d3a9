#!/bin/bash
# SQ counters of the step's kernels (one rocprofv3 --pmc pass over bench.py; nothing but --pmc is combined with it):
# MFMA-busy cycles, wave cycles split into parked / issue-stalled / issuing, LDS activity and bank conflicts.
# usage (GPU box): tools/pmc_sq.sh <tag> [bench.py arguments]   -> gpurun_out/pmc_<tag>_sq/ + gpurun_out/pmc_sq_<tag>.md
cd /tmp && export TMPDIR=/tmp
tag=${1:-r01}; shift
d=/root/repo/gpurun_out/pmc_${tag}_sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $d -o b -- \
    python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer "$@" > $d.log 2>&1
tail -1 $d.log | cut -c1-200
python /root/repo/tools/pmc_sq.py $d/b_counter_collection.csv > /root/repo/gpurun_out/pmc_sq_${tag}.md
head -30 /root/repo/gpurun_out/pmc_sq_${tag}.md
