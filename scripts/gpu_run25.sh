#!/bin/bash
mkdir -p gpurun_out
timeout 600 python scripts/bench_multiscale.py 2>&1 | tail -1 | tee gpurun_out/bench_multiscale.json
timeout 600 python bench.py --arch resnet50 --size 224 --batch 64 --steps 50 --warmup 5 --cpu-seconds 0 2>/dev/null | tail -1 | cut -c1-330 | tee gpurun_out/bench_cfgA.json
timeout 600 python bench.py --dtype fp16 --cpu-seconds 0 2>/dev/null | tail -1 | cut -c1-200 | tee gpurun_out/bench_fp16.json
