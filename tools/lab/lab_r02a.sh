#!/bin/bash
# round-2 lab call A: fused-MLP kernel tests + timing + SQ counters (GPU box)
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "mlp_fused" 2>&1 | tail -15 > gpurun_out/r02a_test.log
cat gpurun_out/r02a_test.log | tail -5
timeout 200 python tools/mlp_lab.py > gpurun_out/r02a_lab.jsonl 2>&1
cat gpurun_out/r02a_lab.jsonl
cd /tmp && export TMPDIR=/tmp
d=/root/repo/gpurun_out/pmc_r02a_sq
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $d -o b -- \
    python /root/repo/tools/mlp_lab.py > $d.log 2>&1
python /root/repo/tools/pmc_sq.py $d/b_counter_collection.csv > /root/repo/gpurun_out/pmc_sq_r02a.md
cat /root/repo/gpurun_out/pmc_sq_r02a.md
rm -rf $d
