#!/bin/bash
mkdir -p gpurun_out
for fmt in u8 f32; do
  timeout 300 python scripts/bench_host_feed.py --fmt $fmt 2>&1 | tail -1 | tee -a gpurun_out/host_feed.txt
done
