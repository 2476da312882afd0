#!/usr/bin/env python
"""Copies the summaries of one evidence run (scripts/gpu/r6_final.sh <tag>: validate.sh + workloads.sh + the world-size-1
launches) from gpurun_out/<tag>/ into profiles/ under the round's names:
    python scripts/collect_evidence.py r6fin3 r06 "<one line: which commit / box>"
-> profiles/rNN_bench_b32_kernel_roofline.txt, _kernel_stats.txt, _layers.txt, rNN_bench_line.json, rNN_workloads.json,
   rNN_gpu_tests_summary.txt and profiles/traffic.json (stamped with the kernel-source hash bench.py checks)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
what = sys.argv[3] if len(sys.argv) > 3 else ''
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


for a, b in (('kernel_roofline.txt', '%s_bench_b32_kernel_roofline.txt'), ('kernel_stats.txt', '%s_bench_b32_kernel_stats.txt'),
             ('bench_layers.txt', '%s_bench_b32_layers.txt')):
    shutil.copy(os.path.join(src, a), os.path.join(dst, b % rnd))
shutil.copy(os.path.join(src, 'traffic.json'), os.path.join(dst, 'traffic.json'))
json.dump(last_json(os.path.join(src, 'bench.json')), open(os.path.join(dst, '%s_bench_line.json' % rnd), 'w'), indent=1)
work = {}
for k in ('distractors', 'multiscale', 'cfgA'):
    work[k] = last_json(os.path.join(src, 'work', k + '.json'))
work['batch1'] = json.load(open(os.path.join(src, 'work', 'batch1.json')))
for k in ('ws1_extract', 'ws1_extract_cal'):
    p = os.path.join(src, k + '.json')
    if os.path.exists(p):
        work[k] = last_json(p)
work['_what'] = 'scripts/gpu/r6_final.sh %s: workloads.sh (configs[3], configs[4], config A, batch 1 on 1-6 streams) and bench.py under ' \
                'torch.distributed.run --nproc-per-node 1 on both checkpoints. %s' % (tag, what)
json.dump(work, open(os.path.join(dst, '%s_workloads.json' % rnd), 'w'), indent=1)
keep = re.compile(r'^\[[a-z0-9/_ -]+\]| passed| failed|^FAILED|^ERROR')
lines = [l.rstrip() for l in open(os.path.join(src, 'pytest.log'), errors='replace') if keep.search(l)]
with open(os.path.join(dst, '%s_gpu_tests_summary.txt' % rnd), 'w') as fh:
    fh.write('\n'.join(lines) + '\n# scripts/gpu/r6_final.sh %s (validate.sh): the whole -m gpu suite; same box as profiles/%s_bench_* and %s_workloads.json. %s\n'
             % (tag, rnd, rnd, what))
print('collected', tag, '->', dst)
