#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2s}
mkdir -p $O
timeout 900 python -m pytest tests/test_ranking_gpu.py -x -q -s -k "split or million" 2>&1 | tail -25
timeout 600 python scripts/bench_rank.py 2>&1 | tail -1 | tee $O/bench_rank.json
