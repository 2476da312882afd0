#!/bin/bash
# round 6: the small-map regime after the heuristic took the tuner's lessons (64x128_w2x2 / 64x64_w2x2_s8 rules): batch 1, 2, 4, config A
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6small2}; mkdir -p $O
run() { tag=$1; shift; timeout 300 python bench.py "$@" --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --no-precision --layers > $O/bench_$tag.json 2> $O/layers_$tag.txt; echo $tag $(python -c "import json;d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['step_mfma_frac'])"); }
run b1 --batch 1
run b2 --batch 2
run b4 --batch 4
run cfgA --arch resnet50 --size 224 --batch 64
run cfgA_tuned --arch resnet50 --size 224 --batch 64 --autotune
run ms849 --size 849 --batch 16
timeout 300 python scripts/bench_batch1.py > $O/batch1.json 2> $O/batch1.err; echo "batch1 rc=$?"; cut -c1-700 $O/batch1.json
