#!/bin/bash
# sample GPU clocks while a conv loop runs
python scripts/time_conv.py "256,14,256,256,3,1,0" "72" 250000 > gpurun_out/clk_conv.log 2>&1 &
PID=$!
sleep 7
for i in $(seq 1 12); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|socclk" | tr '\n' ' '; echo; sleep 0.5; done
wait $PID
cat gpurun_out/clk_conv.log | grep -v amdgpu
