#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for b in 8 16 24; do
  timeout 300 python bench.py --workload multiscale --ms-batch $b --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms-batch $b:', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
