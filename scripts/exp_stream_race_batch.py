#!/usr/bin/env python
"""Multi-stream soak of the BATCHED kernel mix (round 5: the loader / consumer form of conv_patch3x3w has a new hand-off structure):
forwards of a batch-8 1024^2 input issued round-robin on 1 / 2 / 3 HIP streams must be bit-identical to the single-stream result."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import synth  # noqa: E402
from dirtorch_amd import nets  # noqa: E402
from dirtorch_amd.test_dir import StreamPool  # noqa: E402

net = nets.create_model('resnet101_rmac', pretrained='')
net.load_state_dict(synth.synth_state_dict('resnet101', seed=7))
net.compute_dtype = 'fp16p'
net.cuda().eval()
B = int(os.environ.get('RACE_B', 8))
g = torch.Generator(device='cuda').manual_seed(3)
HW = tuple(int(v) for v in os.environ.get('RACE_HW', '1024x1024').split('x'))   # (round 6, late: RACE_B=1 RACE_HW=683x1024 puts conv_small.hip's tile on every layer3 conv)
imgs = [torch.randint(0, 256, (B, HW[0], HW[1], 3), generator=g, dtype=torch.uint8, device='cuda') for _ in range(3)]
net.set_profiling(True)
refs = [net(x).clone() for x in imgs]
kernels = sorted({r['kernel'] for r in net.get_profile()})
net.set_profiling(False)
torch.cuda.synchronize()
print('kernel mix:', [k for k in kernels if 'patch3x3w' in k or 'patchs2' in k or 'c3c1' in k or 'wreg' in k or 'small' in k])
for ns in ((1, 2, 4, 6) if B == 1 else (1, 2, 3)):
    pool = StreamPool(ns)
    outs = []
    for i in range(int(os.environ.get('RACE_N', 36))):
        x = imgs[i % 3]
        outs.append(pool.run(lambda: net(x), x))
    pool.join()
    torch.cuda.synchronize()
    bad = sum(int(not torch.equal(o, refs[i % 3])) for i, o in enumerate(outs))
    print('batch %d, %d streams: %d of %d forwards differ from the single-stream descriptors' % (B, ns, bad, len(outs)))
