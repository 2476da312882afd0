#!/bin/bash
# paired head, third pass: residual prefetch in conv_pair, the patch-pair 3x3 kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4g; mkdir -p $O
timeout 1200 python -m pytest tests/test_pair_gpu.py -q -s --maxfail=8 > $O/pair_tests.log 2>&1; echo "pair tests rc=$?"
grep -E "^\[fp16p\]|passed|failed|^FAILED|^E  " $O/pair_tests.log | head -30
timeout 300 python bench.py --dtype fp16p --layers --cpu-seconds 0 --steps 12 --profile-every 100 > $O/bench_fp16p.json 2> $O/bench_fp16p_layers.txt; echo "bench fp16p rc=$?"
head -c 300 $O/bench_fp16p.json; echo
grep -E "conv_pair|stem_pool|prep_input|layer2.0.conv1" $O/bench_fp16p_layers.txt
DIRTORCH_AMD_NO_PAIR_PATCH=1 timeout 300 python bench.py --dtype fp16p --layers --cpu-seconds 0 --steps 12 --profile-every 100 2>&1 >/dev/null | grep -E "conv2 .*conv_pair"
