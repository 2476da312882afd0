#!/bin/bash
O=gpurun_out/${1:-r5batch}; mkdir -p $O
for B in 32 40 48 56 64 32; do
  timeout 300 python bench.py --batch $B --steps 20 --warmup 3 --cpu-seconds 0 > $O/bench_b$B.json 2> $O/err_b$B.txt
  echo "batch $B: $(python -c "import json;d=json.loads(open('$O/bench_b$B.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
done
