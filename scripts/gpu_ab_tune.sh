#!/bin/bash
# parity of every variant, then A/B on one box: committed tuned table vs fresh autotune
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -x > gpurun_out/ops20.log 2>&1; echo "exit $?" >> gpurun_out/ops20.log; tail -2 gpurun_out/ops20.log
pick='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["all_conv_ms_per_step"])'
cp profiles/r01_tuned_variants_b32_1024.txt /tmp/old_tune.txt
rm -f /tmp/new_tune.txt
for i in 1 2 3; do
  echo -n "old "; DIRTORCH_AMD_TUNE_CACHE=/tmp/old_tune.txt timeout 600 python bench.py --cpu-seconds 0 --autotune 2>/dev/null | tail -1 | python -c "$pick"
  echo -n "new "; DIRTORCH_AMD_TUNE_CACHE=/tmp/new_tune.txt timeout 600 python bench.py --cpu-seconds 0 --autotune 2>/dev/null | tail -1 | python -c "$pick"
done
cp /tmp/new_tune.txt gpurun_out/tune_b32_v4.txt
awk '{print $3}' gpurun_out/tune_b32_v4.txt | sort | uniq -c | sort -rn | head -8
