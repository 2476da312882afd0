#!/usr/bin/env python
"""Which op misbehaves when forwards overlap on several HIP streams?  Every conv shape of a batch-1 ResNet-101 forward at
1024^2 (tile variant and split-K factor as the engine picks them), the fused stem, the input conversion, the two-source
GEMMs and the pooling run as independent jobs: first one at a time (reference), then shuffled over N streams so that
different kernels overlap on the device; every output is compared bit for bit with its reference."""
import os
import random
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from dirtorch_amd import ops  # noqa: E402

dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == 'fp16') else torch.bfloat16
NS = int(os.environ.get('RACE_STREAMS', '6'))
REPS = int(os.environ.get('RACE_REPS', '6'))
g = torch.Generator(device='cuda').manual_seed(5)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g, device='cuda') * scale).to(dt)


jobs = {}


def conv_job(name, H, Cin, Cout, k, stride, res):
    pad = 1 if k == 3 else 0
    OH = (H + 2 * pad - k) // stride + 1
    x = rnd(1, H, H, Cin).relu_()
    w = rnd(Cout, k, k, Cin, scale=(2.0 / (k * k * Cin)) ** 0.5)
    b = torch.randn(Cout, generator=g, device='cuda') * 0.1
    r = rnd(1, OH, OH, Cout) if res else None
    jobs[name] = lambda: ops.conv_bn_act(x, w, b, r, stride=stride, pad=pad, relu=True, ksplit=-1)


# (H of the input map, Cin, Cout, k, stride, residual)
for name, s in {
        'l1.0.conv1': (256, 64, 64, 1, 1, 0), 'l1.conv1': (256, 256, 64, 1, 1, 0), 'l1.conv2': (256, 64, 64, 3, 1, 0),
        'l1.conv3': (256, 64, 256, 1, 1, 1), 'l1.0.ds': (256, 64, 256, 1, 1, 0),
        'l2.0.conv1': (256, 256, 128, 1, 1, 0), 'l2.0.conv2': (256, 128, 128, 3, 2, 0), 'l2.conv1': (128, 512, 128, 1, 1, 0),
        'l2.conv2': (128, 128, 128, 3, 1, 0), 'l2.conv3': (128, 128, 512, 1, 1, 1),
        'l3.0.conv1': (128, 512, 256, 1, 1, 0), 'l3.0.conv2': (128, 256, 256, 3, 2, 0), 'l3.conv1': (64, 1024, 256, 1, 1, 0),
        'l3.conv2': (64, 256, 256, 3, 1, 0), 'l3.conv3': (64, 256, 1024, 1, 1, 1),
        'l4.0.conv1': (64, 1024, 512, 1, 1, 0), 'l4.0.conv2': (64, 512, 512, 3, 2, 0), 'l4.conv1': (32, 2048, 512, 1, 1, 0),
        'l4.conv2': (32, 512, 512, 3, 1, 0), 'l4.conv3': (32, 512, 2048, 1, 1, 1)}.items():
    conv_job(name, *s)
for name, (H, P, Cx) in {'l2.0.dual': (128, 128, 256), 'l3.0.dual': (64, 256, 512), 'l4.0.dual': (32, 512, 1024)}.items():
    t2, xx = rnd(1, H, H, P).relu_(), rnd(1, 2 * H, 2 * H, Cx).relu_()
    wcat = rnd(4 * P, P + Cx, scale=(2.0 / (P + Cx)) ** 0.5)
    bb = torch.randn(4 * P, generator=g, device='cuda') * 0.1
    jobs[name] = (lambda t2=t2, xx=xx, wcat=wcat, bb=bb: ops.conv_dual(t2, xx, wcat, bb, stride2=2, relu=True))
img = torch.randint(0, 256, (1, 1024, 1024, 3), generator=g, dtype=torch.uint8, device='cuda')
jobs['prep_input'] = lambda: ops.prep_input(img, dtype=dt)
s2d = ops.prep_input(img, dtype=dt)
wst = ops.pack_stem_weight(torch.randn(64, 3, 7, 7, generator=g, device='cuda') * 0.1, dtype=dt)
bst = torch.randn(64, generator=g, device='cuda') * 0.1
jobs['stem_pool'] = lambda: ops.stem_pool(s2d, wst, bst, (512, 512))
feat = rnd(1, 32, 32, 2048).relu_()
jobs['global_pool'] = lambda: ops.global_pool(feat, 'gem', 3.0)

refs = {}
for n, f in jobs.items():
    try:
        refs[n] = f().clone()
    except Exception as e:   # an op the C ABI does not take in this form: say so and drop it
        print('skipped %s: %s' % (n, str(e)[:100]))
torch.cuda.synchronize()
names = sorted(refs)
streams = [torch.cuda.Stream() for _ in range(NS)]
bad = {n: 0 for n in names}
total = {n: 0 for n in names}
rng = random.Random(1)
for rep in range(REPS):
    order = names * 4
    rng.shuffle(order)
    outs = []
    for i, n in enumerate(order):
        with torch.cuda.stream(streams[i % NS]):
            outs.append((n, jobs[n]()))
    torch.cuda.synchronize()
    # one comparison kernel per output, one host sync per repetition
    flags = torch.stack([(o != refs[n]).any() for n, o in outs]).cpu().tolist()
    for (n, o), f in zip(outs, flags):
        total[n] += 1
        bad[n] += int(f)
print('dtype %s, %d streams, %d launches: ' % (dt, NS, sum(total.values())) +
      (', '.join('%s %d/%d' % (n, bad[n], total[n]) for n in names if bad[n]) or 'every output identical to its single-stream reference'))
