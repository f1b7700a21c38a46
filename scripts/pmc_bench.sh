#!/bin/bash
# HBM traffic of one bench step from PMC counters (run on the GPU box): separate passes for FETCH_SIZE and
# WRITE_SIZE (TCC has 4 slots; FETCH_SIZE costs 3, WRITE_SIZE 2), kernel-trace only.  bench.py runs WITHOUT its batch-2 check forward
# (--no-check), so every profiled launch is a full-batch one; forwards = warmup 1 + 2 x steps 3 = 7 (timed region + event-bracket pass).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcb_$C
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmcb_$C -- python3 bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --preheat 0 --no-check --dump-layers gpurun_out/pmcb_layers.json > gpurun_out/pmcb_$C.log 2>&1 || echo "pass $C failed"
done
python3 scripts/pmc_bench_summary.py --forwards 7 --layers gpurun_out/pmcb_layers.json --table gpurun_out/pmc_hbm_per_launch.txt > gpurun_out/pmc_hbm_traffic.json 2> gpurun_out/pmc_hbm_dropped.txt
