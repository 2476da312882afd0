#!/bin/bash
# Round-3 experiment 5: loader / consumer ring kernel - tests, standalone timing with phases compiled out, bench A/B.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3h
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -k "ring" -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 $O/pytest.log | cut -c1-400
export EXP_SHAPES=l3.conv1,l4.conv1,l3.0.conv1
V="128x256_ring1x1 256x256_persist1x1 256x256_persist1x1_x3"
python scripts/exp_conv_time.py $V 2>&1 | grep -v "^lib\|amdgpu" | sed 's/^/full        /' | tee $O/abl.txt
for bits in 4 8; do
  DIRTORCH_AMD_LIB=scripts/_exp/libdir_ring$bits.so python scripts/exp_conv_time.py 128x256_ring1x1 2>&1 | grep -v "^lib\|amdgpu" | sed "s/^/abl $bits       /" | tee -a $O/abl.txt
done
unset EXP_SHAPES
B="python bench.py --cpu-seconds 0 --steps 30 --warmup 5"
for rep in 1 2; do
  DIRTORCH_AMD_NO_RING=1 $B > $O/ab_base_$rep.json 2>/dev/null
  $B > $O/ab_ring_$rep.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3h/ab_*.json')):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d['value'], d['ms_per_step'], [(r[0],r[1],r[3]) for r in d['roofline']['kernels']['rows'] if 'conv1' in r[1] and 'layer' in r[1]])
    except Exception as e: print(f, 'ERR', e)
P
