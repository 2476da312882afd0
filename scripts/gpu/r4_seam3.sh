#!/bin/bash
# round 4: layer3 seam kernel (conv_seam3.hip) - op tests, then A/B inside the network
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "layer3_seam or fused_seam_argument" > $O/seam3_tests.log 2>&1; echo "seam3 tests rc=$?"
tail -n 25 $O/seam3_tests.log
for env in "A=1" "DIRTORCH_AMD_NO_SEAM3=1"; do
  echo "== $env"
  env $env timeout 300 python bench.py --cpu-seconds 0 --steps 12 --layers > $O/bench_$env.json 2> $O/layers_$env.txt
  python -c "
import json,sys
d=json.load(open('$O/bench_$env.json')); r=d['roofline']; print(d['ms_per_step'], r['all_kernels_ms_per_step'], d['value'])"
  grep -E "layer3\.(5|6)\." $O/layers_$env.txt
done
