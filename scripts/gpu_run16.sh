#!/bin/bash
# occupancy-2/3 k32 variants: parity of every variant, then autotuned bench with the per-layer table
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x > gpurun_out/k32_tests.log 2>&1
echo "exit $?" >> gpurun_out/k32_tests.log
tail -3 gpurun_out/k32_tests.log
rm -f gpurun_out/tune_b32_v2.txt
DIRTORCH_AMD_TUNE_CACHE=gpurun_out/tune_b32_v2.txt timeout 600 python bench.py --steps 12 --warmup 2 --cpu-seconds 0 --layers > gpurun_out/bench_k32.json 2> gpurun_out/bench_k32_layers.txt
tail -1 gpurun_out/bench_k32.json | cut -c1-300
awk '{print $3}' gpurun_out/tune_b32_v2.txt | sort | uniq -c | sort -rn
