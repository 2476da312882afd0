#!/bin/bash
# the fp16-pair similarity kernel + the lock-step rank histogram: tests, then the distractor workload both ways
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_ranking_gpu.py -q -x 2>&1 | tail -n 3
for mode in pair general; do
  if [ $mode = general ]; then X=--sim-general; else X=; fi
  timeout 600 python bench.py --workload distractors --steps 10 --warmup 3 --cpu-seconds 0 $X 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$mode', d['ms_per_step'], 'ms/step; sim', r['avg_launch_ms'], 'ms', r['achieved'], 'GB/s frac', r['frac'], r['kernel'], 'rank+ap', r['rank_ap_ms'], 'mAP', d['config']['mAP_medium'])"
done
