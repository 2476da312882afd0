#!/bin/bash
# round 6: the small-map regime with the deep-ring small-tile variants (64x64_w2x2_s8, 64x128_w2x2_s6, 128x64_w2x2_s6): parity of the
# new table entries, then batch 1 and config A with the heuristic's picks and with the autotuner's (per-layer tables of both)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6small}; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "s8 or s6" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
run() { tag=$1; shift; timeout 300 python bench.py "$@" --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --no-precision --layers > $O/bench_$tag.json 2> $O/layers_$tag.txt; echo $tag $(python -c "import json;d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['step_mfma_frac'])"); }
run b1 --batch 1
run b1_tuned --batch 1 --autotune
run cfgA --arch resnet50 --size 224 --batch 64
run cfgA_tuned --arch resnet50 --size 224 --batch 64 --autotune
run b4_tuned --batch 4 --autotune
run b4 --batch 4
