#!/bin/bash
# PMC passes over one GEMM shape (separate rocprofv3 runs per counter set; --pmc must not be combined with tracing).
# usage: tools/pmc_gemm.sh <tag> nt|tn P Q K
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_LEVEL_WAVES SQ_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /root/repo/gpurun_out/pmc_${tag}_$i -o g -- python /root/repo/tools/gemm_lab.py "$@" 3 > /root/repo/gpurun_out/pmc_${tag}_$i.log 2>&1
  grep TFLOP /root/repo/gpurun_out/pmc_${tag}_$i.log
done
python /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/pmc_${tag}_*/g_counter_collection.csv
