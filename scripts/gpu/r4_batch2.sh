#!/bin/bash
# round 4: graph replay, sharded scoring, the default bench line with config.workloads
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4f; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ranking_gpu.py tests/test_comm_gpu.py -q -x -k "graph_replay or sharded_scoring or rccl or allgather" 2>&1 | tail -n 15
timeout 600 python scripts/bench_batch1.py > $O/batch1.json 2> $O/batch1.err; echo "batch1 rc=$?"; python -c "
import json; d=json.load(open('$O/batch1.json'))
for k,v in d.items(): print(k, v)"
tail -3 $O/batch1.err
timeout 900 python bench.py --steps 10 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['config']['workloads'])); print(json.dumps(d['config']['precision']['images_per_sec']), json.dumps(d['config']['precision']['one_minus_cos']))"
tail -3 $O/bench.err
