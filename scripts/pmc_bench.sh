#!/bin/bash
# HBM traffic of one bench step from PMC counters (run on the GPU box): separate passes for FETCH_SIZE and
# WRITE_SIZE (TCC has 4 slots; FETCH_SIZE costs 3, WRITE_SIZE 2), kernel-trace only.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmcb_$C -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --preheat 0 > gpurun_out/pmcb_$C.log 2>&1 || echo "pass $C failed"
done
