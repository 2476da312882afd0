#!/bin/bash
# Round-3 experiment 10: the two-source (conv3 + downsample) form of the split loader / consumer ring kernel.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3p
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -k "two_source or ring or fused_seams" -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-400
B="python bench.py --cpu-seconds 0 --steps 30 --warmup 5"
for rep in 1 2; do
  DIRTORCH_AMD_DUAL_RING=0 $B > $O/ab_base_$rep.json 2>/dev/null
  $B > $O/ab_ring_$rep.json 2>/dev/null
done
DIRTORCH_AMD_DUAL_RING=force $B > $O/ab_force_1.json 2>/dev/null
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3p/ab_*.json')):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d['value'], d['ms_per_step'], [(r[0],r[1],r[3]) for r in d['roofline']['kernels']['rows'] if 'dual' in r[0]])
    except Exception as e: print(f, 'ERR', e)
P
