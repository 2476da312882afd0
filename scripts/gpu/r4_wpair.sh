#!/bin/bash
# DIR_FP16P with paired 1x1 WEIGHTS in layer1 (conv_c3c1.hip WP3 / WP1): the pair tests, then the step in both forms.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4wp
timeout 1200 python -m pytest tests/test_pair_gpu.py -q -x -s 2>&1 | grep -E "^\[fp16p|passed|failed|Error|error|assert" | cut -c1-230 | tail -n 30
for form in weights acts; do
  if [ $form = acts ]; then export DIRTORCH_AMD_PAIR_ACTS=1; else unset DIRTORCH_AMD_PAIR_ACTS; fi
  timeout 600 python bench.py --dtype fp16p --steps 30 --warmup 5 --no-workloads --no-precision --cpu-seconds 0 --layers > gpurun_out/r4wp/bench_$form.json 2> gpurun_out/r4wp/layers_$form.txt
  python - <<PY
import json
d = json.loads(open('gpurun_out/r4wp/bench_$form.json').read().strip().splitlines()[-1])
print('$form', d['value'], d['unit'], d['ms_per_step'], 'ms')
PY
  head -n 16 gpurun_out/r4wp/layers_$form.txt | cut -c1-170
done
