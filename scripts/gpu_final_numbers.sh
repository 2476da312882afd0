pick='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "ms/step", d["ms_per_step"])'
for b in 1 2 4 8 16; do echo -n "B=$b: "; timeout 600 python bench.py --cpu-seconds 0 --batch $b --steps $((b<8?100:30)) --warmup 5 --profile-every 1000 2>/dev/null | tail -1 | python -c "$pick"; done
echo -n "fp16 B=32: "; timeout 600 python bench.py --cpu-seconds 0 --dtype fp16 2>/dev/null | tail -1 | python -c "$pick"
echo -n "cfgA R50@224 B=64: "; timeout 600 python bench.py --arch resnet50 --size 224 --batch 64 --steps 50 --warmup 5 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "$pick"
timeout 600 python scripts/bench_multiscale.py 2>&1 | tail -1
