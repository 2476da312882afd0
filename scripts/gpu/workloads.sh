#!/bin/bash
# The other workloads of the bench contract: configs[3] distractors, configs[4] multiscale, config A,
# batch 1 on 1-6 streams.   gpurun -- 'bash scripts/gpu/workloads.sh <tag>'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-workloads}
mkdir -p $O
timeout 300 python bench.py --workload distractors --steps 10 --warmup 2 > $O/distractors.json 2> $O/distractors.err; echo "distractors rc=$?"
timeout 300 python bench.py --workload multiscale --steps 8 --warmup 2 > $O/multiscale.json 2> $O/multiscale.err; echo "multiscale rc=$?"
timeout 300 python scripts/bench_batch1.py > $O/batch1.json 2> $O/batch1.err; echo "batch1 rc=$?"
timeout 300 python bench.py --arch resnet50 --size 224 --batch 64 --cpu-seconds 0 --no-precision > $O/cfgA.json 2> $O/cfgA.err; echo "cfgA rc=$?"
O=$O python - <<'P'
import json, os
O = os.environ['O']
for f in ('distractors', 'multiscale', 'cfgA'):
    try:
        d = json.load(open('%s/%s.json' % (O, f))); r = d['roofline']
        print(f, d['value'], d['unit'], d['ms_per_step'], 'ms/step', r.get('kernel'), r.get('frac'), r.get('avg_launch_ms'), r.get('rank_ap_ms'), d.get('cpu_baseline'))
    except Exception as e:
        print(f, 'ERR', e)
print(open('%s/batch1.json' % O).read()[:1200])
P
