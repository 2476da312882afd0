#!/usr/bin/env python
"""configs[4]'s step (three scales of a 16 x 1200^2 uint8 batch, fp16p) with the three forwards issued on ONE stream and on the
host mirror's stream pool (dirtorch_amd.test_dir.StreamPool: what extract_multiscale_features does for its images): do the
under-filled kernels of the small scale and the tail rounds of the large one fill each other's idle CUs?"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import synth
from dirtorch_amd import nets, ops
from dirtorch_amd.test_dir import StreamPool
from dirtorch_amd.utils import common, transforms

B, S = int(os.environ.get('EXP_BATCH', 16)), 1200
sd = synth.synth_state_dict('resnet101', seed=7)
net = nets.create_model('resnet101_rmac', pretrained='')
net.load_state_dict(sd)
net.compute_dtype = 'fp16p'
net = net.cuda().eval()
g = torch.Generator(device='cuda').manual_seed(99)
img = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device='cuda', generator=g)
scales = [transforms.Scale(0.7071), None, transforms.Scale(1.4142)]
sizes = [(S, S) if sc is None else sc.target_size((S, S)) for sc in scales]


def step_seq():
    per = []
    for size in sizes:
        x = img if size == (S, S) else ops.resize_bilinear_u8(img, size)
        per.append(net(x))
    return common.l2_normalize(common.pool(per, 'gem', 3))


def make_pool(n):
    os.environ['DIRTORCH_AMD_STREAMS'] = str(n)
    return StreamPool()


def step_pool(pool, order):
    per = [None] * 3
    for i in order:
        size = sizes[i]
        x = img if size == (S, S) else ops.resize_bilinear_u8(img, size)
        per[i] = pool.run(lambda: net(x), x)
    pool.join()
    return common.l2_normalize(common.pool(per, 'gem', 3))


def rate(fn, steps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


ref = step_seq()
print('one stream            %.1f three-scale img/s' % rate(step_seq))
for n in (2, 3):
    pool = make_pool(n)
    for order in ((0, 1, 2), (2, 1, 0), (2, 0, 1)):
        out = step_pool(pool, order)
        print('%d streams, order %s  %.1f img/s   bit-identical to one stream: %s' % (n, order, rate(lambda: step_pool(pool, order)), torch.equal(out, ref)))
