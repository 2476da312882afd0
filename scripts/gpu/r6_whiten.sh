#!/bin/bash
# Round 6: whitening at database scale (sim_split.hip whiten_split_kernel) - parity tests, then configs[3]'s line with whiten_ms
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6whiten}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_whiten_gpu.py -m gpu -q -s -p no:cacheprovider > $O/pytest_whiten.log 2>&1; echo "pytest whiten rc=$?"
grep -a "^\[whiten\| passed\| failed\|^FAILED\|^ERROR\|Error\|assert" $O/pytest_whiten.log | tail -30
timeout 600 python bench.py --workload distractors --steps 10 --warmup 2 --cpu-seconds 0 > $O/distractors.json 2> $O/distractors.err; echo "distractors rc=$?"
python - <<P
import json
d = json.loads(open('$O/distractors.json').read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], 'ms/step; whiten:', json.dumps(d['roofline'].get('whiten')))
P
tail -3 $O/distractors.err
