#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest28.log 2>&1; echo "exit $?" >> gpurun_out/pytest28.log
grep -v "Warning\|pin_memory\|^$" gpurun_out/pytest28.log | tail -12
pick='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "ms/step", d["ms_per_step"])'
for b in 1 2 4 8 32; do echo -n "B=$b: "; timeout 600 python bench.py --cpu-seconds 0 --batch $b --steps $((b<8?100:20)) --warmup 5 --profile-every 1000 2>/dev/null | tail -1 | python -c "$pick"; done
echo -n "B=1 heuristic: "; timeout 600 python bench.py --cpu-seconds 0 --batch 1 --steps 100 --warmup 5 --profile-every 1000 --no-autotune 2>/dev/null | tail -1 | python -c "$pick"
