#!/bin/bash
# Second SQ pass over bench.py (VERDICT round 3, item 1 "day one"): what the LDS and the VALU of the step's kernels actually do -
# LDS-array cycles (SQ_LDS_IDX_ACTIVE) and the bank-conflict cycles among them, LDS / VALU instruction counts and issue stalls.
# Nothing but --pmc is combined with it (8 SQ slots + GRBM).
# usage (GPU box): tools/pmc_lds.sh <tag> [bench.py arguments]  -> gpurun_out/pmc_lds_<tag>.md
cd /tmp && export TMPDIR=/tmp
tag=${1:-r04}; shift
d=/root/repo/gpurun_out/pmc_${tag}_lds
timeout 600 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS \
    SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $d -o b -- \
    python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer "$@" > $d.log 2>&1
tail -1 $d.log | cut -c1-200
python /root/repo/tools/pmc_lds.py $d/b_counter_collection.csv > /root/repo/gpurun_out/pmc_lds_${tag}.md
head -24 /root/repo/gpurun_out/pmc_lds_${tag}.md
rm -rf $d
