#!/bin/bash
# the whole GPU suite, as the driver runs it at round end
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-full}; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -s > $O/gpu_tests.log 2>&1; echo "rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" $O/gpu_tests.log | tail -n 30
