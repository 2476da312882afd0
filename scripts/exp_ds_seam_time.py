#!/usr/bin/env python
"""Standalone timing of layer1's DS seam (conv3 + downsample + next conv1, dir_conv_c3c1_ds[_wpair]) at batch 32 of 1024^2:
the loader / consumer form (conv_c3c1lc.hip) against the one-role kernel (DIRTORCH_AMD_NO_C3C1LC=1), paired and plain."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-image-retrieval_amd'))
import torch
from dirtorch_amd import _lib, ops

B, H, W = int(os.environ.get('EXP_BATCH', 32)), 256, 256
g = torch.Generator(device='cuda').manual_seed(1)
h = lambda *sh, s=1.0: (torch.randn(*sh, generator=g, device='cuda') * s).half()      # noqa: E731
t2, xh, xl = torch.relu(h(B, H, W, 64)), torch.relu(h(B, H, W, 64)), h(B, H, W, 64, s=2.0 ** -11)
wh, wl = h(256, 128, s=0.06), h(256, 128, s=0.06 * 2.0 ** -11)
w1h, w1l = h(64, 256, s=0.04), h(64, 256, s=0.04 * 2.0 ** -11)
b, b1 = torch.randn(256, generator=g, device='cuda'), torch.randn(64, generator=g, device='cuda')
M = B * H * W
forms = {'paired (fp16p)': (lambda: ops.conv_c3c1_ds_wpair(t2, (xh, xl), (wh, wl), b, (w1h, w1l), b1), 2.0 * M * (64 * 3 + 256 + 64)),
         'plain fp16': (lambda: ops.conv_c3c1_ds(t2, xh, wh, b, w1h, b1), 2.0 * M * (64 * 2 + 256 + 64))}
for fname, (fn, byts) in forms.items():
    outs = []
    for arm in ('', 'DIRTORCH_AMD_NO_C3C1LC=1'):
        if arm:
            os.environ['DIRTORCH_AMD_NO_C3C1LC'] = '1'
        _lib.reload_env()
        y, t1 = fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        os.environ.pop('DIRTORCH_AMD_NO_C3C1LC', None)
        _lib.reload_env()
        outs.append((y, t1))
        print('%-16s %-28s %.4f ms  %6.0f GB/s' % (fname, arm or 'roles split (default)', best, byts / best / 1e6))
    print('%-16s bit-identical: %s' % (fname, torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])))
