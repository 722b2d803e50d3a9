#!/bin/bash
# SQ counters of a lab command (GPU box): tools/lab_pmc.sh <tag> <command...>
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
d=/root/repo/gpurun_out/pmc_${tag}_sq
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $d -o b -- "$@" > $d.log 2>&1
python /root/repo/tools/pmc_sq.py $d/b_counter_collection.csv > /root/repo/gpurun_out/pmc_sq_${tag}.md
cat /root/repo/gpurun_out/pmc_sq_${tag}.md
rm -rf $d
