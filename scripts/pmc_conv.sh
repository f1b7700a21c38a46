#!/bin/bash
# usage: scripts/pmc_conv.sh <tag> <one_conv args...>   (run on the GPU box; writes gpurun_out/pmc_<tag>_*)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_BRANCH"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $P --output-format csv -d gpurun_out/pmc_${tag}_p$i -- python3 scripts/one_conv.py "$@" > gpurun_out/pmc_${tag}_p$i.log 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
