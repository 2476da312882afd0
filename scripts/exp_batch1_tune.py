#!/usr/bin/env python
"""Batch 1 at the reference's native sizes (test_dir.py feeds images one at a time) through the autotuner: picker vs tuner per layer."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import synth
from dirtorch_amd import nets

sd = synth.synth_state_dict('resnet101', seed=7)


def engine(tune):
    net = nets.create_model('resnet101_rmac', pretrained='')
    net.load_state_dict(sd)
    net.compute_dtype = 'fp16p'
    net = net.cuda().eval()
    net.autotune = tune
    return net


g = torch.Generator(device='cuda').manual_seed(3)
for (H, W) in ((1024, 1024), (768, 1024), (683, 1024), (1024, 819), (500, 375)):
    x = torch.randint(0, 256, (1, H, W, 3), dtype=torch.uint8, device='cuda', generator=g)
    prof, ms = {}, {}
    for tune in (False, True):
        net = engine(tune)
        with torch.no_grad():
            net(x)
            net.autotune = False
            for _ in range(3):
                net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                net(x)
            torch.cuda.synchronize()
            ms[tune] = (time.perf_counter() - t0) / 20 * 1e3
            net.set_profiling(True)
            net(x)
            prof[tune] = {r['name']: (r['kernel'], r['ms']) for r in net.get_profile()}
            net.set_profiling(False)
        del net
    print('size %dx%d: picker %.3f ms (%.0f img/s), tuner %.3f ms (%.0f img/s)' % (H, W, ms[False], 1e3 / ms[False], ms[True], 1e3 / ms[True]))
    agg = {}
    for name, (k0, t0_) in prof[False].items():
        k1, t1 = prof[True].get(name, (None, 0.0))
        if k1 is not None and k1 != k0:
            parts = name.split('.')
            key = ('%s.%s' % (parts[0], parts[-1]) if len(parts) == 3 and parts[1] != '0' else name, k0, k1)
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += t0_
            a[2] += t1
    for (name, k0, k1), (n, a, b) in sorted(agg.items(), key=lambda kv: kv[1][2] - kv[1][1])[:8]:
        print('   %-16s x%-2d %-36s %.3f -> %-36s %.3f ms' % (name, n, k0, a, k1, b))
