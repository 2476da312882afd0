#!/bin/bash
# Round-3 experiment 7: staggered LDS-DMA issue in the 3x3 kernel (conv_patchw.hip STAG) - tests + A/B; multiscale workload.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3j
mkdir -p $O
DIRTORCH_AMD_PATCHW_STAG=1 timeout 600 python -m pytest tests/test_ops_gpu.py -k "3x3" -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-300
B="python bench.py --cpu-seconds 0 --steps 30 --warmup 5"
for rep in 1 2 3; do
  $B > $O/ab_base_$rep.json 2>/dev/null
  DIRTORCH_AMD_PATCHW_STAG=1 $B > $O/ab_stag_$rep.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3j/ab_*.json')):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d['value'], d['ms_per_step'], [(r[1],r[3]) for r in d['roofline']['kernels']['rows'] if 'patch3x3w' in r[0]])
    except Exception as e: print(f, 'ERR', e)
P
timeout 300 python bench.py --workload multiscale --steps 8 --warmup 2 > $O/multiscale.json 2> $O/multiscale.err; echo "multiscale rc=$?"; cut -c1-900 $O/multiscale.json; tail -2 $O/multiscale.err
