#!/usr/bin/env python
"""Timing of the paired stem (dir_stem_pool_pair: conv 7x7 s2 + BN + ReLU + max-pool on fp16 pairs, csrc/conv_pair.hip) at the
bench shape (batch 32 at 1024^2), standalone.  With DIRTORCH_AMD_LIB=scripts/_exp/lib_conv_pair_<bits>.so (scripts/exp_abl.sh
conv_pair DIR_STEMP_ABL <bits>) the kernel runs with phases compiled out - timing only."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-image-retrieval_amd'))
import torch
from dirtorch_amd import _lib, ops

B, S = int(os.environ.get('EXP_B', 32)), 1024
g = torch.Generator(device='cuda').manual_seed(1)
img = torch.randn(B, 3, S, S, device='cuda', generator=g)
w7 = torch.randn(64, 3, 7, 7, device='cuda', generator=g) * (2.0 / 147) ** 0.5
bias = torch.randn(64, device='cuda', generator=g) * 0.1
s2d = ops.prep_input_pair(img)
wp = ops.split_pair(ops.pack_stem_weight(w7, torch.float32))
OH = (S + 6 - 7) // 2 + 1


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


t = timeit(lambda: ops.stem_pool_pair(s2d, wp, bias, (OH, OH)))
tp = timeit(lambda: ops.prep_input_pair(img))
print('lib %s | stem_pool_pair %.1f us | prep_input_pair %.1f us' % (os.path.basename(_lib.LIB_PATH), t, tp))
