#!/usr/bin/env python
"""Statistics for the in-suite guard (tests/test_model_gpu.py::test_forwards_overlapping_on_streams_are_bit_identical[fp16p]):
the uint8 stem (csrc/stem_u8.hip - counted waits, LDS exchange buffers, register-prefetched patches) under heavy stream
overlap: N forwards of a ResNet-18 engine (the stem is most of its time) on 4 streams against the single-stream result."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import synth
from dirtorch_amd import nets
from dirtorch_amd.test_dir import StreamPool

net = nets.create_model('resnet18_rmac', pretrained='')
net.load_state_dict(synth.synth_state_dict('resnet18', seed=7))
net.compute_dtype = 'fp16p'
net.cuda().eval()
g = torch.Generator(device='cuda').manual_seed(3)
bad = total = 0
for (H, W) in ((512, 640), (1024, 1024), (333, 500), (767, 1023)):
    imgs = [torch.randint(0, 256, (1, H, W, 3), generator=g, dtype=torch.uint8, device='cuda') for _ in range(4)]
    refs = [net(x).clone() for x in imgs]
    torch.cuda.synchronize()
    pool = StreamPool(4)
    for rep in range(int(os.environ.get('EXP_REPS', 40))):
        outs = [pool.run(lambda x=imgs[i % 4]: net(x), imgs[i % 4]) for i in range(64)]
        pool.join()
        torch.cuda.synchronize()
        bad += sum(not torch.equal(o, refs[i % 4]) for i, o in enumerate(outs))
        total += len(outs)
    print('%dx%d: %d of %d overlapped forwards differ from the single-stream descriptors so far' % (H, W, bad, total), flush=True)
