#!/bin/bash
mkdir -p gpurun_out
DIRTORCH_AMD_TUNE_CACHE=gpurun_out/tune_b1.txt timeout 600 python bench.py --cpu-seconds 0 --batch 1 --steps 50 --warmup 10 --profile-every 1 --layers > gpurun_out/b1.json 2> gpurun_out/b1_layers.txt
sort -k4 -n -r gpurun_out/b1_layers.txt | head -5
awk '{split($1,a,"."); key=a[1]; if (a[3]!="") key=a[1]"."a[3]; t[key]+=$3; n[key]++} END{for(k in t) printf "%-22s %3d launches %7.3f ms\n", k, n[k], t[k]}' gpurun_out/b1_layers.txt | sort -k4 -n -r
