#!/usr/bin/env python
"""configs[4]'s three scales through the autotuner: per scale (one stream, batch 16 of 1200^2 resized), the per-layer kernels and
times of the picker's choice against the tuner's - which of the odd map sizes (54^2 / 75^2 / 107^2 / 38^2 / 27^2 ...) would rather
run on a flattened implicit-GEMM tile than on the 16 x 32-pixel patch tile?"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import synth
from dirtorch_amd import nets, ops
from dirtorch_amd.utils import transforms

B, S = int(os.environ.get('EXP_BATCH', 16)), 1200
sd = synth.synth_state_dict('resnet101', seed=7)


def engine(tune):
    net = nets.create_model('resnet101_rmac', pretrained='')
    net.load_state_dict(sd)
    net.compute_dtype = 'fp16p'
    net = net.cuda().eval()
    net.autotune = tune
    return net


g = torch.Generator(device='cuda').manual_seed(99)
img = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device='cuda', generator=g)
scales = [transforms.Scale(0.7071), None, transforms.Scale(1.4142)]
sizes = [(S, S) if sc is None else sc.target_size((S, S)) for sc in scales]
nets_ = {False: engine(False), True: engine(True)}
for size in sizes:
    x = img if size == (S, S) else ops.resize_bilinear_u8(img, size)
    prof, ms = {}, {}
    for tune, net in nets_.items():
        with torch.no_grad():
            net(x)                       # (tunes this size's layer shapes when asked to)
            net.autotune = False
            net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                net(x)
            torch.cuda.synchronize()
            ms[tune] = (time.perf_counter() - t0) / 5 * 1e3
            net.set_profiling(True)
            net(x)
            prof[tune] = {r['name']: (r['kernel'], r['ms']) for r in net.get_profile()}
            net.set_profiling(False)
            net.autotune = tune
    print('size %s: picker %.3f ms, tuner %.3f ms' % (size, ms[False], ms[True]))
    agg = {}
    for name, (k0, t0_) in prof[False].items():
        k1, t1 = prof[True].get(name, (None, 0.0))
        if k1 is not None and k1 != k0:
            key = (name.split('.')[0] + '.' + name.split('.')[-1] if name.count('.') == 2 and name.split('.')[1] not in ('0',) else name, k0, k1)
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += t0_
            a[2] += t1
    for (name, k0, k1), (n, a, b) in sorted(agg.items(), key=lambda kv: kv[1][2] - kv[1][1]):
        print('   %-16s x%-2d %-36s %.3f -> %-36s %.3f ms' % (name, n, k0, a, k1, b))
