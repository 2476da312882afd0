#!/usr/bin/env python
"""The headline step (32 x 1024^2 uint8, ResNet-101 GeM, fp16p) as ONE forward of 32 images against k forwards of 32 / k images
on k streams of the host mirror's pool: do concurrent sub-batches fill each other's kernel tails?  Measured (round 6): no -
2 462 img/s as one forward, 2 468 as 2 x 16 on two streams, 2 326 as 2 x 16 on one, 2 083-2 167 as 4 x 8."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import synth
from dirtorch_amd import nets
from dirtorch_amd.test_dir import StreamPool

B, S = 32, 1024
sd = synth.synth_state_dict('resnet101', seed=7)
net = nets.create_model('resnet101_rmac', pretrained='')
net.load_state_dict(sd)
net.compute_dtype = 'fp16p'
net = net.cuda().eval()
g = torch.Generator(device='cuda').manual_seed(5)
img = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device='cuda', generator=g)


def rate(fn, steps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


ref = net(img)
print('one forward of 32            %.1f img/s' % rate(lambda: net(img)))
for k, n in ((2, 2), (4, 2), (4, 4), (2, 1)):
    parts = [img[i * (B // k):(i + 1) * (B // k)].contiguous() for i in range(k)]
    pool = StreamPool(n)

    def step():
        outs = [pool.run(lambda x=x: net(x), x) for x in parts]
        pool.join()
        return torch.cat(outs)
    out = step()
    print('%d forwards of %2d on %d stream(s)  %.1f img/s   max |d| vs one forward %.3g' % (k, B // k, n, rate(step), float((out - ref).abs().max())))
