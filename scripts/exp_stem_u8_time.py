#!/usr/bin/env python
"""Timing of the uint8-feed stem (csrc/stem_u8.hip: prep_input_u8 + stem_pool_u8) at the bench shape (batch 32 at 1024^2) through
the engine's own profile records (a ResNet-18 engine: the stem is the same, the rest of the forward is short).  With
DIRTORCH_AMD_LIB=scripts/_exp/lib_stem_u8_<bits>.so (scripts/exp_abl.sh stem_u8 DIR_STEMU8_ABL <bits>) the kernel runs with phases
compiled out - timing only.  DIRTORCH_AMD_STEM_U8_SEG=T / DIRTORCH_AMD_NO_STEM_U8=1 select the segment length / the generic paired stem."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import synth
from dirtorch_amd import _lib, nets

B, S = int(os.environ.get('EXP_B', 32)), int(os.environ.get('EXP_S', 1024))
net = nets.create_model('resnet18_rmac', pretrained='')
net.load_state_dict(synth.synth_state_dict('resnet18', seed=7))
net.compute_dtype = 'fp16p'
net.cuda().eval()
g = torch.Generator(device='cuda').manual_seed(1)
if os.environ.get('EXP_PICS'):
    sys.path.insert(0, ROOT)
    import bench
    pics = bench.to_uint8_nhwc(synth.synth_images(1234, 8, S, S)).cuda()
    img = torch.stack([pics[i % 8] for i in range(B)])
else:
    img = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device='cuda', generator=g)
if os.environ.get('EXP_F32'):      # the fp32 NCHW feed: prep_input_pair + the generic paired stem
    sys.path.insert(0, ROOT)
    import bench
    img = bench.normalise_uint8(img.cpu()).cuda()
for _ in range(3):
    net(img)
torch.cuda.synchronize()
net.set_profiling(True)
for _ in range(10):
    net(img)
torch.cuda.synchronize()
rows = {}
for r in net.get_profile():
    if r['name'] in ('prep_input', 'conv1+maxpool'):
        rows.setdefault((r['name'], r['kernel']), []).append(r['ms'])
out = ' | '.join('%s %s %.1f us (min %.1f)' % (k[0], k[1], 1e3 * sum(v) / len(v), 1e3 * min(v)) for k, v in rows.items())
print('lib %s seg=%s | %s' % (os.path.basename(_lib.LIB_PATH), os.environ.get('DIRTORCH_AMD_STEM_U8_SEG', '-'), out))
