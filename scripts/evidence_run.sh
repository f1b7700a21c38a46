#!/bin/bash
# Evidence sweep on the GPU box (outputs under gpurun_out/): batch-size fuzz, accuracy of every precision on both weight families,
# a 2000-step bench (is the 20-step number representative?), the input / PCIe variants of the bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python3 scripts/fuzz_batch.py 2>&1 | grep -v amdgpu > gpurun_out/r4_fuzz.txt; cat gpurun_out/r4_fuzz.txt
(timeout -k 10 300 python3 scripts/accuracy_report.py uniform; timeout -k 10 300 python3 scripts/accuracy_report.py trained) 2>&1 | grep -v amdgpu > gpurun_out/r4_accuracy.txt; cat gpurun_out/r4_accuracy.txt
python3 bench.py --steps 2000 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null > gpurun_out/r4_bench_2000steps.json
python3 -c "
import json; d=json.load(open('gpurun_out/r4_bench_2000steps.json')); print('2000 steps:', d['value'], d['ms_per_step'], d['preheat'])" | tee gpurun_out/r4_bench_inputs.txt
for v in "--input u8" "--input video" "--from-host" "--input u8 --from-host" "--batch 512" "--precision fp8 --batch 512" "--precision fp8 --batch 256"; do
  python3 bench.py $v --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],3))" | tee -a gpurun_out/r4_bench_inputs.txt
done
