#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_heads_gpu.py -q -k "fp16p" 2>&1 | tail -n 3
SKIP_TESTS=1 BENCH_ARGS="--dtype fp16p" bash scripts/gpu/validate.sh r4v2p 2>&1 | tail -n 45 | cut -c1-190
