#!/bin/bash
# round 4, call 2: paired head with single-plane block outputs, fused downsample, persistent stem
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b; mkdir -p $O
timeout 1200 python -m pytest tests/test_pair_gpu.py -q -s --maxfail=8 > $O/pair_tests.log 2>&1; echo "pair tests rc=$?"
grep -E "^\[fp16p|passed|failed|^FAILED|^E  " $O/pair_tests.log | head -40
timeout 300 python bench.py --dtype fp16p --layers --cpu-seconds 0 --steps 8 > $O/bench_fp16p.json 2> $O/bench_fp16p_layers.txt; echo "bench fp16p rc=$?"
head -c 330 $O/bench_fp16p.json; echo
grep -E "conv_pair|stem_pool|prep_input" $O/bench_fp16p_layers.txt
DIRTORCH_AMD_STEM_V1=1 timeout 300 python bench.py --dtype fp16p --layers --cpu-seconds 0 --steps 8 2>&1 >/dev/null | grep -E "stem_pool"
