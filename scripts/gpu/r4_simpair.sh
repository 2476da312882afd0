#!/bin/bash
# the fp16-pair similarity kernel: tests, then the distractor workload with it and with the six-product kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_ranking_gpu.py -q -x -s -k "similarity" 2>&1 | grep -E "^\[|passed|failed|^E " | cut -c1-200 | tail -n 24
for mode in pair general; do
  if [ $mode = general ]; then X=--sim-general; else X=; fi
  timeout 600 python bench.py --workload distractors --steps 10 --warmup 3 --cpu-seconds 0 $X 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$mode', d['ms_per_step'], 'ms/step; sim', r['avg_launch_ms'], 'ms', r['achieved'], 'GB/s frac', r['frac'], r['kernel'], 'rank+ap', r['rank_ap_ms'], 'mAP', d['config']['mAP_medium'])"
done
