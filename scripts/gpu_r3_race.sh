#!/bin/bash
# After ring_barrier(): op-level and forward-level multi-stream reproducibility, op tests, headline bench.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3m
mkdir -p $O
for q in 4 8; do GPU_MAX_HW_QUEUES=$q RACE_REPS=1200 RACE_STREAMS=6 timeout 300 python scripts/exp_stream_race_ops.py 2>&1 | grep -v amdgpu | cut -c1-300; done
RACE_SIZES=1024x1024,768x1024 RACE_STREAMS=4,6 RACE_REPS=40 GPU_MAX_HW_QUEUES=8 python scripts/exp_stream_race.py 2>&1 | grep -v amdgpu | cut -c1-200
timeout 600 python -m pytest tests/test_ops_gpu.py -k "3x3 or ring or persist or wreg or stem" -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for rep in 1 2; do python bench.py --cpu-seconds 0 --steps 30 --warmup 5 > $O/bench_$rep.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_$rep.json')); print('bench', d['value'], d['ms_per_step'])"; done
python scripts/bench_batch1.py 2>/dev/null | cut -c1-700
