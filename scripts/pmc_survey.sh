#!/bin/bash
# Where do the wave cycles of every kernel of a bench step go?  Three rocprofv3 --pmc passes over bench.py (3 steps), summarised per kernel
# by scripts/pmc_survey_summary.py.  Run on the GPU box; writes gpurun_out/pmcs_p{1,2,3} and gpurun_out/pmc_survey.txt.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_BRANCH"
i=1
for P in "$P1" "$P2" "$P3"; do
  rm -rf gpurun_out/pmcs_p$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d gpurun_out/pmcs_p$i -- python3 bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --preheat 0 > gpurun_out/pmcs_p$i.log 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
python3 scripts/pmc_survey_summary.py > gpurun_out/pmc_survey.txt
