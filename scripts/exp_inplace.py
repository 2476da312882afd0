#!/usr/bin/env python
"""Round-5 probe: do the identity blocks of layers 3-4 run faster when a sub-batch's tensors fit the
256 MiB Infinity Cache?  Runs the bench network (R101 @ 1024^2, fp16p) at batch 8 / 16 / 32 with the block outputs
ping-ponging (DIRTORCH_AMD_NO_INPLACE=1) and IN PLACE (the default since the end of round 5: map + t1 + t2 = 12.6 MB per image in layer3 -> 201 MB at
batch 16), prints the layer3 / layer4 per-launch times of both forms and checks the descriptors are bit-identical."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import synth  # noqa: E402
from dirtorch_amd import _lib, nets  # noqa: E402


def run(B, inplace, steps=12):
    if inplace:          # (the default since the end of round 5)
        os.environ.pop('DIRTORCH_AMD_NO_INPLACE', None)
    else:
        os.environ['DIRTORCH_AMD_NO_INPLACE'] = '1'
    _lib.load()
    _lib.reload_env()
    net = nets.create_model('resnet101_rmac', pretrained='')
    net.load_state_dict(synth.synth_state_dict('resnet101', seed=7))
    net.compute_dtype = 'fp16p'
    net.cuda().eval()
    g = torch.Generator(device='cuda').manual_seed(1234)
    x = torch.randn(B, 3, 1024, 1024, generator=g, device='cuda')
    for _ in range(3):
        d = net(x)
    net.set_profiling(256 * (steps + 2))
    for _ in range(steps):
        d = net(x)
    torch.cuda.synchronize()
    prof = net.get_profile()
    net.set_profiling(False)
    agg = {}
    for r in prof:
        parts = r['name'].split('.')
        key = (parts[0] + '.' + parts[-1]) if (len(parts) == 3 and parts[1] != '0') else r['name']
        a = agg.setdefault(key, [0.0, 0])
        a[0] += r['ms']
        a[1] += 1
    total = sum(r['ms'] for r in prof) / steps
    return d.clone(), {k: v[0] / v[1] * 1e3 for k, v in agg.items()}, total


out = {}
for B in (8, 16, 32):
    d0, t0, s0 = run(B, False)
    d1, t1, s1 = run(B, True)
    same = bool(torch.equal(d0, d1))
    row = {'descriptors_bit_identical': same, 'step_ms': [round(s0, 3), round(s1, 3)],
           'us_per_image_layer3_block': [round((t0['layer3.conv1'] + t0['layer3.conv2'] + t0['layer3.conv3']) / B, 3),
                                         round((t1['layer3.conv1'] + t1['layer3.conv2'] + t1['layer3.conv3']) / B, 3)]}
    for k in ('layer3.conv1', 'layer3.conv2', 'layer3.conv3', 'layer4.conv1', 'layer4.conv2', 'layer4.conv3'):
        row[k + '_us'] = [round(t0[k], 1), round(t1[k], 1)]
    out['batch_%d' % B] = row
    print('batch %2d  ping-pong | in place:  step %.3f | %.3f ms   layer3 block %.2f | %.2f us/img   conv1 %.1f | %.1f  conv2 %.1f | %.1f  '
          'conv3 %.1f | %.1f us   layer4 conv1 %.1f | %.1f conv3 %.1f | %.1f   identical: %s' % (
              B, s0, s1, row['us_per_image_layer3_block'][0], row['us_per_image_layer3_block'][1],
              t0['layer3.conv1'], t1['layer3.conv1'], t0['layer3.conv2'], t1['layer3.conv2'], t0['layer3.conv3'], t1['layer3.conv3'],
              t0['layer4.conv1'], t1['layer4.conv1'], t0['layer4.conv3'], t1['layer4.conv3'], same), flush=True)
print(json.dumps(out))
