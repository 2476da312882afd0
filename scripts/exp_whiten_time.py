#!/usr/bin/env python
"""Timing of dir_pca_whiten_l2_unit (csrc/sim_split.hip whiten_split_kernel) on N unit-norm 2048-d rows, v = 2048 components;
under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE the HBM bytes of the launch say whether the workgroups that share an X tile meet in L2."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-image-retrieval_amd'))
import torch
from dirtorch_amd import ops

N, D, v = int(os.environ.get('EXP_N', 262144)), 2048, int(os.environ.get('EXP_V', 2048))
g = torch.Generator(device='cuda').manual_seed(1)
X = torch.nn.functional.normalize(torch.randn(N, D, device='cuda', generator=g).abs_(), dim=1)
mean = X[:8192].mean(dim=0).contiguous()
comps = torch.linalg.qr(torch.randn(D, D, device='cuda', generator=g))[0].t().contiguous()[:v].contiguous()
alpha = torch.ones(v, device='cuda')
for _ in range(2):
    ops.pca_whiten(X, comps, mean, alpha, unit_range=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ms = []
for _ in range(5):
    e0.record()
    out = ops.pca_whiten(X, comps, mean, alpha, unit_range=True)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
    del out
ms.sort()
fl = 2.0 * N * D * v
print('whiten %d x %d x %d: %.3f ms (min %.3f) = %.1f TFLOP/s algorithmic; X is %.2f GB, out %.2f GB' % (
    N, D, v, ms[2], ms[0], fl / ms[2] / 1e9, N * D * 4 / 1e9, N * v * 4 / 1e9))
