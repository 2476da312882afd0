#!/bin/bash
# conv3 + downsample on the persistent two-source ring (conv_persist.hip DUAL) against conv_igemm.hip's tile: tests, then the
# three first-block lines of the layer profile, both ways.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4dual
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "two_source" 2>&1 | tail -n 4
for form in persist igemm; do
  if [ $form = igemm ]; then export DIRTORCH_AMD_DUAL_IGEMM=1; else unset DIRTORCH_AMD_DUAL_IGEMM; fi
  timeout 600 python bench.py --steps 30 --warmup 5 --no-workloads --no-precision --cpu-seconds 0 --layers > gpurun_out/r4dual/bench_$form.json 2> gpurun_out/r4dual/layers_$form.txt
  python - <<PY
import json
d = json.loads(open('gpurun_out/r4dual/bench_$form.json').read().strip().splitlines()[-1])
print('$form', d['value'], d['unit'], d['ms_per_step'], 'ms')
PY
  grep -E "ds\+conv3" gpurun_out/r4dual/layers_$form.txt | cut -c1-150
done
