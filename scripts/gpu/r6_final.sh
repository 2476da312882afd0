#!/bin/bash
# Round 6 evidence run: the whole -m gpu suite, the default bench line, the rocprofv3 passes behind profiles/r06_*, the other workloads,
# and the multi-rank path at world size 1.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r6fin}
bash scripts/gpu/validate.sh $TAG
bash scripts/gpu/workloads.sh $TAG/work
O=gpurun_out/$TAG
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29631"
timeout 600 $TR bench.py --gpus 1 --steps 20 --warmup 3 --cpu-seconds 0 > $O/ws1_extract.json 2> $O/ws1_extract.err; echo "ws1 extract rc=$?"
timeout 600 $TR bench.py --gpus 1 --steps 20 --warmup 3 --cpu-seconds 0 --checkpoint calibrated > $O/ws1_extract_cal.json 2> $O/ws1_extract_cal.err; echo "ws1 extract (calibrated: rank 0 calibrates) rc=$?"
python - <<P
import json
for f in ('ws1_extract', 'ws1_extract_cal'):
    try:
        d = json.loads(open('$O/%s.json' % f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], 'rccl_ranks', d['config']['rccl_ranks'], d['config']['timed_checkpoint'][:30])
    except Exception as e:
        print(f, 'ERR', e)
P
