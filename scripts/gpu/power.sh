#!/bin/bash
# sustained shader clock and power while the bench loop runs
mkdir -p gpurun_out
(DIRTORCH_AMD_TUNE_CACHE=/tmp/t.txt python bench.py --steps 2500 --warmup 3 --cpu-seconds 0 > gpurun_out/clk_bench.json 2>/dev/null) &
BP=$!
sleep 22
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power (W)\|Average Graphics\|Socket" | tr '\n' ' '; echo
  sleep 1
done | tee gpurun_out/clk.txt
wait $BP
tail -1 gpurun_out/clk_bench.json | cut -c1-200
