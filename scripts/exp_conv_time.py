#!/usr/bin/env python
"""Per-variant timing of single conv launches on the layer3/4 shapes (B = 32 at 1024^2).
With DIRTORCH_AMD_LIB=scripts/_exp/libdir_fill.so the igemm variants run their LDS-DMA ring and
epilogue but no MFMAs, which separates 'fill-bound' from 'MFMA-bound'."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-image-retrieval_amd'))
import ctypes
import torch
from dirtorch_amd import _lib, ops

SHAPES = {   # name: (B, H, W, Cin, Cout, k, stride, pad, residual)
    'l3.conv1': (32, 64, 64, 1024, 256, 1, 1, 0, False),
    'l3.conv2': (32, 64, 64, 256, 256, 3, 1, 1, False),
    'l3.conv3': (32, 64, 64, 256, 1024, 1, 1, 0, True),
    'l2.conv2': (32, 128, 128, 128, 128, 3, 1, 1, False),
    'l2.conv3': (32, 128, 128, 128, 512, 1, 1, 0, True),
    'l4.conv2': (32, 32, 32, 512, 512, 3, 1, 1, False),
    'l4.conv1': (32, 32, 32, 2048, 512, 1, 1, 0, False),
    'l3.0.conv1': (32, 128, 128, 512, 256, 1, 1, 0, False),
    'l2.0.conv2': (32, 256, 256, 128, 128, 3, 2, 1, False), 'l3.0.conv2': (32, 128, 128, 256, 256, 3, 2, 1, False),
    'l4.0.conv2': (32, 64, 64, 512, 512, 3, 2, 1, False),   # the strided 3x3 convs of the first blocks (round 6: conv_patchs2.hip)
}
names = []
n = _lib.load().dir_conv_variant_count()
for v in range(n):
    buf = ctypes.create_string_buffer(64)
    _lib.call('dir_conv_variant_name', v, buf, 64)
    names.append(buf.value.decode())
ZEROS = os.environ.get('EXP_ZEROS') == '1'   # zero-filled operands: same instruction stream, far less switching power
want = sys.argv[1:] or ['256x256_w4x2', '256x256_w4x4', '256x256_w4x2_s3_k32', '256x256_w4x2_s4_k32',
                        '256x256_persist1x1', '128x256_w2x4_s3_k32', '256x128_w4x2_s3_k32', '128x128_w2x2']
print('lib', _lib.LIB_PATH)
ONLY = os.environ.get('EXP_SHAPES')   # comma-separated subset of SHAPES
for sname, (B, H, W, Cin, Cout, k, st, pad, res) in SHAPES.items():
    if ONLY and sname not in ONLY.split(','):
        continue
    x = (torch.randn(B, H, W, Cin, device='cuda') * 0.5).to(torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, device='cuda') * 0.02).to(torch.bfloat16)
    if ZEROS:
        x.zero_()
        w.zero_()
    bias = torch.zeros(Cout, device='cuda')
    OH = (H + 2 * pad - k) // st + 1
    r = (torch.randn(B, OH, OH, Cout, device='cuda')).to(torch.bfloat16) if res else None
    flops = 2.0 * B * OH * OH * Cout * k * k * Cin
    byts = 2.0 * (x.numel() + B * OH * OH * Cout * (2 if res else 1) + w.numel())
    row = []
    for vn in want:
        if vn not in names:
            continue
        v = names.index(vn)
        try:
            ops.conv_bn_act(x, w, bias, r, st, pad, True, variant=v)
        except Exception:
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(5):
                ops.conv_bn_act(x, w, bias, r, st, pad, True, variant=v)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        row.append('%s %.3f' % (vn, best))
    print('%-9s %6.1f GF %6.0f MB | ' % (sname, flops / 1e9, byts / 1e6) + ' | '.join(row))
