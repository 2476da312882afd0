#!/usr/bin/env python
"""Timing of the layer3 seam (dir_conv_c3c1 at planes 256, conv_seam3.hip) against the two launches it replaces, on
the bench shape (batch 32 at 1024^2: M = 131072 pixels).  With DIRTORCH_AMD_LIB=scripts/_exp/lib_conv_seam3_<bits>.so
(scripts/exp_abl.sh) the fused kernel runs with phases compiled out - timing only."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-image-retrieval_amd'))
import torch
from dirtorch_amd import _lib, ops

B, H, W, P = int(os.environ.get('EXP_B', 32)), 64, 64, 256
dt = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(1)
t2 = torch.relu(torch.randn(B, H, W, P, device='cuda', generator=g)).to(dt)
res = torch.relu(torch.randn(B, H, W, 4 * P, device='cuda', generator=g)).to(dt)
w3 = (torch.randn(4 * P, 1, 1, P, device='cuda', generator=g) * (2.0 / P) ** 0.5).to(dt)
w1 = (torch.randn(P, 1, 1, 4 * P, device='cuda', generator=g) * (0.5 / P) ** 0.5).to(dt)
b3, b1 = torch.zeros(4 * P, device='cuda'), torch.zeros(P, device='cuda')


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


fused = timeit(lambda: ops.conv_c3c1(t2, w3, b3, res, w1, b1))
line = 'lib %s | fused %.1f us' % (os.path.basename(_lib.LIB_PATH), fused)
if not os.environ.get('DIRTORCH_AMD_LIB'):
    y = ops.conv_bn_act(t2, w3, b3, res, relu=True)
    c3 = timeit(lambda: ops.conv_bn_act(t2, w3, b3, res, relu=True))
    c1 = timeit(lambda: ops.conv_bn_act(y, w1, b1, None, relu=True))
    line += ' | conv3 %.1f + conv1 %.1f = %.1f us' % (c3, c1, c3 + c1)
print(line)
