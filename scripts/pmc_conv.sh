#!/bin/bash
# usage: scripts/pmc_conv.sh <tag> <one_conv args...>   (run on the GPU box; writes gpurun_out/pmc_<tag>_*)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
P3="SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS TA_BUSY_avr"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $P --output-format csv -d gpurun_out/pmc_${tag}_p$i -- python3 scripts/one_conv.py "$@" > gpurun_out/pmc_${tag}_p$i.log 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
