#!/bin/bash
# Round-2 side numbers for BASELINE.md: small batches, fp16, config A / D / E, power + clock under load.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2n}
mkdir -p $O
pick='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms/step")'
for b in 1 2 4 8 16; do echo -n "B=$b: "; timeout 600 python bench.py --cpu-seconds 0 --batch $b --steps $((b<8?100:30)) --warmup 5 --profile-every 1000 2>/dev/null | tail -1 | python -c "$pick"; done | tee $O/small_batch.txt
echo -n "fp16 B=32: " | tee -a $O/small_batch.txt; timeout 600 python bench.py --cpu-seconds 0 --dtype fp16 2>/dev/null | tail -1 | python -c "$pick" | tee -a $O/small_batch.txt
echo -n "cfgA R50@224 B=64: " | tee -a $O/small_batch.txt; timeout 600 python bench.py --arch resnet50 --size 224 --batch 64 --steps 50 --warmup 5 --cpu-seconds 8 2>/dev/null | tail -1 > $O/cfgA.json; python -c "$pick" < $O/cfgA.json | tee -a $O/small_batch.txt
timeout 600 python scripts/bench_multiscale.py 2>&1 | tail -1 | tee $O/multiscale.json
timeout 600 python scripts/bench_rank.py 2>&1 | tail -1 | tee $O/rank.json
# power / clock while the bench loop runs
(python bench.py --steps 2500 --warmup 3 --cpu-seconds 0 > $O/clk_bench.json 2>/dev/null) &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power (W)\|Average Graphics\|Socket" | tr '\n' ' '; echo
  sleep 1
done | tee $O/power_clock.txt
wait $BP
tail -1 $O/clk_bench.json | cut -c1-160
