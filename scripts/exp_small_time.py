#!/usr/bin/env python
"""Per-variant timing of single conv launches on the SMALL-MAP shapes (batch 1 at 1024^2, config A = ResNet-50 at 224^2
x 64): which tile / ring depth / split-K a layer with 1 000 - 13 000 output pixels wants.  For every shape: the
heuristic's own pick (variant -1, ksplit -1) and each named variant plain and with split-K 2 / 4.
    python scripts/exp_small_time.py [variant ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-image-retrieval_amd'))
import torch
from dirtorch_amd import _lib, ops

SHAPES = {   # name: (B, H, W, Cin, Cout, k, stride, pad, residual)
    'b1.l2.conv1': (1, 128, 128, 512, 128, 1, 1, 0, False),
    'b1.l2.conv2': (1, 128, 128, 128, 128, 3, 1, 1, False),
    'b1.l2.conv3': (1, 128, 128, 128, 512, 1, 1, 0, True),
    'b1.l3.conv1': (1, 64, 64, 1024, 256, 1, 1, 0, False),
    'b1.l3.conv2': (1, 64, 64, 256, 256, 3, 1, 1, False),
    'b1.l3.conv3': (1, 64, 64, 256, 1024, 1, 1, 0, True),
    'b1.l4.conv1': (1, 32, 32, 2048, 512, 1, 1, 0, False),
    'b1.l4.conv2': (1, 32, 32, 512, 512, 3, 1, 1, False),
    'b1.l4.conv3': (1, 32, 32, 512, 2048, 1, 1, 0, True),
    'A.l2.conv2': (64, 28, 28, 128, 128, 3, 1, 1, False),
    'A.l3.conv1': (64, 14, 14, 1024, 256, 1, 1, 0, False),
    'A.l3.conv2': (64, 14, 14, 256, 256, 3, 1, 1, False),
    'A.l3.conv3': (64, 14, 14, 256, 1024, 1, 1, 0, True),
    'A.l4.conv1': (64, 7, 7, 2048, 512, 1, 1, 0, False),
    'A.l4.conv2': (64, 7, 7, 512, 512, 3, 1, 1, False),
    'A.l4.conv3': (64, 7, 7, 512, 2048, 1, 1, 0, True),
}
names = ops.conv_variant_names()
want = sys.argv[1:] or ['64x64_w2x2_s8', '64x128_w2x2_s6', '128x64_w2x2_s6', '64x128_w2x2_s4', '64x128_w2x2', '128x128_w2x2']
DT = torch.float16
ONLY = os.environ.get('EXP_SHAPES')


def time_it(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    return best


for sname, (B, H, W, Cin, Cout, k, st, pad, res) in SHAPES.items():
    if ONLY and not any(sname.startswith(p) for p in ONLY.split(',')):
        continue
    g = torch.Generator(device='cuda').manual_seed(2)
    x = torch.relu(torch.randn(B, H, W, Cin, device='cuda', generator=g)).to(DT)
    w = (torch.randn(Cout, k, k, Cin, device='cuda', generator=g) * (2.0 / (k * k * Cin)) ** 0.5).to(DT)
    bias = torch.randn(Cout, device='cuda', generator=g) * 0.1
    OH = (H + 2 * pad - k) // st + 1
    r = torch.randn(B, OH, OH, Cout, device='cuda', generator=g).to(DT) if res else None
    flops = 2.0 * B * OH * OH * Cout * k * k * Cin
    ref = ops.conv_bn_act(x, w, bias, r, st, pad, True, variant=-1, ksplit=-1)
    t = time_it(lambda: ops.conv_bn_act(x, w, bias, r, st, pad, True, variant=-1, ksplit=-1))
    row = ['pick(k%d) %.1f' % (ops.conv_bn_act.last_ksplit, t * 1e3)]
    for vn in want:
        if vn not in names:
            continue
        v = names.index(vn)
        cell = []
        for ks in (None, 2, 4):
            try:
                y = ops.conv_bn_act(x, w, bias, r, st, pad, True, variant=v, ksplit=ks)
            except Exception:
                continue
            err = float((y.float() - ref.float()).abs().max() / (ref.float().abs().max() + 1e-9))
            t = time_it(lambda: ops.conv_bn_act(x, w, bias, r, st, pad, True, variant=v, ksplit=ks))
            cell.append('%s%.1f%s' % ('' if ks is None else 'k%d:' % ks, t * 1e3, '' if err < 4e-3 else '(ERR %.2g)' % err))
        row.append('%s %s' % (vn, ' '.join(cell)))
    print('%-12s %6.2f GF | ' % (sname, flops / 1e9) + ' | '.join(row) + '   [us]')
