#!/bin/bash
# Round-3 experiment 9: layer1's 3x3 conv with the filter resident in LDS + loader / consumer waves (conv_patchlc.hip).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3n
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -k "3x3_s1-  or 3x3_patch64 or 3x3_multi" -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.log | cut -c1-400
EXP_SHAPES=l1.conv2 python - <<'P'
import os, sys
sys.path.insert(0, 'deep-image-retrieval_amd')
import torch
from dirtorch_amd import ops
names = ops.conv_variant_names()
x = (torch.randn(32, 256, 256, 64, device='cuda') * 0.5).relu_().to(torch.bfloat16)
w = (torch.randn(64, 3, 3, 64, device='cuda') * 0.06).to(torch.bfloat16)
b = torch.randn(64, device='cuda') * 0.1
outs = {}
for vn in ('256x64_patch3x3', '256x64_patchlc3x3'):
    v = names.index(vn)
    y = ops.conv_bn_act(x, w, b, None, 1, 1, True, variant=v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(5):
            ops.conv_bn_act(x, w, b, None, 1, 1, True, variant=v)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    outs[vn] = y
    print('%s: %.3f ms' % (vn, best))
print('bit-identical:', torch.equal(outs['256x64_patch3x3'], outs['256x64_patchlc3x3']))
P
B="python bench.py --cpu-seconds 0 --steps 30 --warmup 5"
for rep in 1 2; do
  DIRTORCH_AMD_NO_PATCHLC=1 $B > $O/ab_base_$rep.json 2>/dev/null
  $B > $O/ab_lc_$rep.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3n/ab_*.json')):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d['value'], d['ms_per_step'], [(r[0],r[1],r[3]) for r in d['roofline']['kernels']['rows'] if 'conv2' in r[1] and 'layer1' in r[1]])
    except Exception as e: print(f, 'ERR', e)
P
