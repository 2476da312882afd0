#!/usr/bin/env python
"""Picker vs tuner per layer for any (arch, batch, H, W): python scripts/exp_tune_any.py resnet50:64:224:224 resnet101:8:1024:1024 ..."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import synth
from dirtorch_amd import nets


def engine(arch, tune):
    net = nets.create_model(arch + '_rmac', pretrained='')
    net.load_state_dict(synth.synth_state_dict(arch, seed=7))
    net.compute_dtype = 'fp16p'
    net = net.cuda().eval()
    net.autotune = tune
    return net


g = torch.Generator(device='cuda').manual_seed(3)
for spec in sys.argv[1:]:
    arch, B, H, W = spec.split(':')
    B, H, W = int(B), int(H), int(W)
    x = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device='cuda', generator=g)
    prof, ms = {}, {}
    for tune in (False, True):
        net = engine(arch, tune)
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
    print('%s batch %d %dx%d: picker %.3f ms (%.0f img/s), tuner %.3f ms (%.0f img/s)' % (arch, B, H, W, ms[False], B * 1e3 / ms[False], ms[True], B * 1e3 / ms[True]))
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
    for (name, k0, k1), (n, a, b) in sorted(agg.items(), key=lambda kv: kv[1][2] - kv[1][1])[:7]:
        print('   %-18s x%-2d %-36s %.3f -> %-36s %.3f ms' % (name, n, k0, a, k1, b))
