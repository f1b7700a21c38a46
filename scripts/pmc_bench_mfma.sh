#!/bin/bash
# MFMA-pipe utilisation per kernel class of a bench step from PMC counters (run on the GPU box), kernel-trace only; full-batch launches only.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcb_MFMA
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmcb_MFMA -- python3 bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --preheat 0 --no-check > gpurun_out/pmcb_MFMA.log 2>&1 || echo "pass failed"
python3 scripts/pmc_bench_mfma_summary.py --forwards 7 > gpurun_out/pmc_mfma_util.json
