#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2k}
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ranking_gpu.py -m gpu -q -s -p no:cacheprovider -k "gemm or million or similarity or widths or expand or named" > $O/pytest_new.log 2>&1
echo "pytest(new) rc=$?"
grep -a " passed\| failed\|^FAILED\|^ERROR\|Error\|^E  " $O/pytest_new.log | tail -12
echo "--- old kernel"; DIRTORCH_AMD_NO_WIDE_GEMM=1 timeout 600 python scripts/bench_rank.py 2>&1 | tail -1 | tee $O/rank_old.json
echo "--- wide kernel"; timeout 600 python scripts/bench_rank.py 2>&1 | tail -1 | tee $O/rank.json
bash scripts/gpu_r2_numbers.sh ${1:-r2k}
