#!/bin/bash
# round 6: conv_small.hip (64 x 64 tiles, loader / consumer waves on LDS counters): parity of the two table entries (every shape of
# the op-level matrix, under a short timeout: a lost hand-off must show up as a failed test, not as a hung box), per-shape timing at
# batch 1 / config A, then batch 1 with the tuner's picks
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6convsmall}; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "small_s" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python scripts/exp_small_time.py 64x64_small_s8 64x64_small_s4 64x64_w2x2_s4 64x128_w2x2_s4 128x128_w2x2 2>&1 | grep -v amdgpu.ids | tee $O/small_time.txt
run() { tag=$1; shift; timeout 300 python bench.py "$@" --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --no-precision --layers > $O/bench_$tag.json 2> $O/layers_$tag.txt; echo $tag $(python -c "import json;d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"); }
run b1 --batch 1
run b1_tuned --batch 1 --autotune
run cfgA_tuned --arch resnet50 --size 224 --batch 64 --autotune
