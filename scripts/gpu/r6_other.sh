#!/bin/bash
# round 6: per-layer profiles of the OTHER configurations in the headline format (config A, the three scales of configs[4], batch 1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6other}; mkdir -p $O
run() { tag=$1; shift; timeout 300 python bench.py "$@" --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --no-precision --layers > $O/bench_$tag.json 2> $O/layers_$tag.txt; echo $tag $(python -c "import json;d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['step_mfma_frac'])"); }
run cfgA --arch resnet50 --size 224 --batch 64
run ms849 --size 849 --batch 16
run ms1200 --size 1200 --batch 16
run ms1697 --size 1697 --batch 16
run b1 --batch 1
