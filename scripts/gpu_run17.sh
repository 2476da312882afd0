#!/bin/bash
# repeatability: one autotune, then 4 benches reusing its choices
mkdir -p gpurun_out
rm -f gpurun_out/tune_rep.txt
for i in 1 2 3 4; do
  DIRTORCH_AMD_TUNE_CACHE=gpurun_out/tune_rep.txt timeout 600 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['all_conv_ms_per_step'])"
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
