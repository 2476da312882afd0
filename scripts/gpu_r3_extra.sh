#!/bin/bash
# The other workloads of the bench contract + the batch-1 figures, final build of round 3.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3y
mkdir -p $O
timeout 300 python bench.py --workload distractors --steps 10 --warmup 2 > $O/distractors.json 2> $O/distractors.err; echo "distractors rc=$?"
timeout 300 python bench.py --workload multiscale --steps 8 --warmup 2 > $O/multiscale.json 2> $O/multiscale.err; echo "multiscale rc=$?"
timeout 300 python scripts/bench_batch1.py > $O/batch1.json 2> $O/batch1.err; echo "batch1 rc=$?"
timeout 300 python bench.py --arch resnet50 --size 224 --batch 64 --cpu-seconds 0 --no-precision > $O/cfgA.json 2> $O/cfgA.err; echo "cfgA rc=$?"
python - <<'P'
import json
for f in ('distractors', 'multiscale', 'cfgA'):
    try:
        d = json.load(open('gpurun_out/r3y/%s.json' % f)); r = d['roofline']
        print(f, d['value'], d['unit'], d['ms_per_step'], 'ms/step', r.get('kernel'), r.get('frac'), r.get('avg_launch_ms'), r.get('rank_ap_ms'), d.get('cpu_baseline'))
    except Exception as e:
        print(f, 'ERR', e)
print(open('gpurun_out/r3y/batch1.json').read()[:1200])
P
