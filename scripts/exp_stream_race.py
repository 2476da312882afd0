#!/usr/bin/env python
"""Do forwards issued on several HIP streams (test_dir.StreamPool) return bit-identical results to the same forwards on
one stream?  Runs the trunk (forward_features) of one batch-1 image many times on 1 and on N streams and compares every
output map bit for bit with the first single-stream result; prints where mismatches sit."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import synth  # noqa: E402
from dirtorch_amd import nets  # noqa: E402
from dirtorch_amd.test_dir import StreamPool  # noqa: E402

arch, dtype = sys.argv[1] if len(sys.argv) > 1 else 'resnet101', sys.argv[2] if len(sys.argv) > 2 else 'fp16'
net = nets.create_model(arch + '_rmac', pretrained='')
net.load_state_dict(synth.synth_state_dict(arch, seed=7))
net.compute_dtype = dtype
net.cuda().eval()
SIZES = [tuple(int(v) for v in t.split('x')) for t in os.environ.get('RACE_SIZES', '1024x1024,768x1024,500x375').split(',')]
NS = [int(v) for v in os.environ.get('RACE_STREAMS', '1,2,4,6').split(',')]
for H, W in SIZES:
    g = torch.Generator(device='cuda').manual_seed(3)
    imgs = [torch.randint(0, 256, (1, H, W, 3), generator=g, dtype=torch.uint8, device='cuda') for _ in range(4)]
    refs = [net.forward_features(x).clone() for x in imgs]
    torch.cuda.synchronize()
    for ns in NS:
        pool = StreamPool(ns)
        bad = 0
        where = []
        for rep in range(int(os.environ.get('RACE_REPS', '3'))):
            outs = []
            for i in range(48):
                x = imgs[i % 4]
                outs.append(pool.run(lambda: net.forward_features(x), x))
            pool.join()
            torch.cuda.synchronize()
            for i, o in enumerate(outs):
                d = (o != refs[i % 4])
                if bool(d.any()):
                    bad += 1
                    if len(where) < 3:
                        idx = d.nonzero()
                        where.append((i, int(d.sum()), idx[0].tolist(), idx[-1].tolist(),
                                      float((o.float() - refs[i % 4].float()).abs().max())))
                    if bad <= 2:   # is it (partly) ANOTHER image's map?  which positions / channels differ?
                        same = [int((o == r).sum()) for r in refs]
                        pos = d[0].any(dim=2)                      # [h, w]: any channel differs
                        chans = d[0].flatten(0, 1).any(dim=0)      # [C]
                        rows = pos.any(dim=1).nonzero().flatten().tolist()
                        cols = pos.any(dim=0).nonzero().flatten().tolist()
                        print('   forward %d (image %d): elements equal to the maps of images 0-3: %s of %d; positions differing %d of %d '
                              '(rows %s..%s, cols %s..%s), channels differing %d of %d, first %s' % (
                                  i, i % 4, same, o.numel(), int(pos.sum()), pos.numel(), rows[:1], rows[-1:], cols[:1], cols[-1:],
                                  int(chans.sum()), chans.numel(), chans.nonzero().flatten()[:12].tolist()))
        print('%s %s %dx%d streams=%d: %d of %d forwards differ from the single-stream map %s' % (
            arch, dtype, H, W, ns, bad, int(os.environ.get('RACE_REPS', '3')) * 48, where[:1]))
