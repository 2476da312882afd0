#!/usr/bin/env python
"""Standalone timing of the two-source GEMMs (conv3 + downsample of the first block of layers 2-4, dir_conv_dual) at
batch 32 of 1024^2, with the environment given on the command line applied per arm:
    python scripts/exp_dual_time.py "" DIRTORCH_AMD_NO_WREGD=1
prints one row per shape and arm (ms, GB/s of algorithmic bytes, TFLOP/s) and checks the arms agree bit for bit."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-image-retrieval_amd'))
import torch
from dirtorch_amd import _lib, ops

SHAPES = {   # name: (B, OH, OW, Cin, Cout, Cin2)
    'l2.0': (32, 128, 128, 128, 512, 256),
    'l3.0': (32, 64, 64, 256, 1024, 512),
    'l4.0': (32, 32, 32, 512, 2048, 1024),
}
arms = sys.argv[1:] or ['']
DT = torch.float16
for sname, (B, OH, OW, Cin, Cout, Cin2) in SHAPES.items():
    if os.environ.get('EXP_SHAPES') and sname not in os.environ['EXP_SHAPES'].split(','):
        continue
    g = torch.Generator(device='cuda').manual_seed(1)
    t2 = torch.relu(torch.randn(B, OH, OW, Cin, device='cuda', generator=g)).to(DT)
    x = torch.relu(torch.randn(B, 2 * OH, 2 * OW, Cin2, device='cuda', generator=g)).to(DT)
    w = (torch.randn(Cout, Cin + Cin2, device='cuda', generator=g) * 0.05).to(DT)
    bias = torch.randn(Cout, device='cuda', generator=g) * 0.1
    M = B * OH * OW
    flops = 2.0 * M * Cout * (Cin + Cin2)
    byts = 2.0 * (M * (Cin + Cin2 + Cout) + Cout * (Cin + Cin2))
    ys = []
    for arm in arms:
        kv = [a.split('=', 1) for a in arm.split() if '=' in a]
        for k, v in kv:
            os.environ[k] = v
        _lib.reload_env()
        y = ops.conv_dual(t2, x, w, bias, stride2=2, relu=True)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv_dual(t2, x, w, bias, stride2=2, relu=True)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        for k, _ in kv:
            del os.environ[k]
        _lib.reload_env()
        ys.append(y)
        print('%-5s %-32s %.4f ms  %6.0f GB/s  %6.0f TF/s' % (sname, arm or '(default)', best, byts / best / 1e6, flops / best / 1e9))
    print('%-5s arms bit-identical: %s' % (sname, all(torch.equal(ys[0], y) for y in ys[1:])))
