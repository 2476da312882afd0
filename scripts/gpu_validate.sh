#!/bin/bash
# full GPU suite + smoke + benches (autotuned x2, default heuristic, batch 1/4)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest18.log 2>&1
echo "exit $?" >> gpurun_out/pytest18.log
grep -v "Warning\|pin_memory\|^$" gpurun_out/pytest18.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
pick='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["all_conv_mfma_frac"])'
for i in 1 2; do timeout 600 python bench.py --cpu-seconds 0 --autotune 2>/dev/null | tail -1 | tee gpurun_out/bench18_$i.json | python -c "$pick"; done
echo heuristic; timeout 600 python bench.py --cpu-seconds 0 2>/dev/null | tail -1 | python -c "$pick"
echo b1; timeout 600 python bench.py --cpu-seconds 0 --batch 1 --steps 200 --warmup 20 --profile-every 1000 2>/dev/null | tail -1 | python -c "$pick"
echo b4; timeout 600 python bench.py --cpu-seconds 0 --batch 4 --steps 100 --warmup 10 --profile-every 1000 2>/dev/null | tail -1 | python -c "$pick"
