#!/bin/bash
# End-of-round refresh on the GPU box: full GPU test suite, default bench line, rocprofv3 kernel stats of the same command, per-layer
# profile, PMC passes (HBM bytes per launch with the per-launch table, MFMA utilisation; full-batch launches only).  Outputs under
# gpurun_out/; copy the summaries into profiles/ (r04_* names in profiles/README.md).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/t_final.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/t_final.log
timeout -k 10 300 python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || echo "bench failed"
rm -rf gpurun_out/prof_final
# kernel stats: ONE lane (the launches of the roofline object: with two lanes in flight kernels of both batches share the chip and a launch's
# duration is not its own), then the default command (two lanes) for the record
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --lanes 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --preheat 0.3 > gpurun_out/prof_final.log 2>&1 || echo "rocprof stats failed"
rm -rf gpurun_out/prof_final2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --preheat 0.3 > gpurun_out/prof_final2.log 2>&1 || echo "rocprof stats (two lanes) failed"
f=$(ls -t gpurun_out/prof_final2/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_final_lanes2.csv
timeout -k 10 300 python3 bench.py --lanes 1 --no-secondary --no-cpu-baseline > gpurun_out/bench_final_lanes1.json 2> gpurun_out/bench_final_lanes1.err || echo "bench (one lane) failed"
timeout -k 10 200 python3 scripts/layer_profile.py > gpurun_out/layer_profile_final.txt 2>&1 || echo "layer profile failed"
bash scripts/pmc_bench.sh
bash scripts/pmc_bench_mfma.sh
f=$(ls -t gpurun_out/prof_final/*/*kernel_trace.csv | head -1); python3 scripts/kernel_gaps.py $f > gpurun_out/kernel_gaps_final.txt 2>&1
f=$(ls -t gpurun_out/prof_final/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_final.csv
timeout -k 10 100 python3 scripts/hbm_copy_probe.py > gpurun_out/hbm_copy_probe.txt 2>&1
tail -3 gpurun_out/t_final.log; cut -c1-300 gpurun_out/bench_final.json
