#!/usr/bin/env python
"""Batch-1 extraction (the reference's real workload, dirtorch/test_dir.py:52-55: one image per forward at its native
size) on one MI355X: images/sec of ResNet-101 at 1024x1024 and 1024x768 with the forwards issued round-robin on
1 ... 6 HIP streams (dirtorch_amd.test_dir.StreamPool - what the extraction loops use).  Prints one JSON line.
(Round 4 also measured hipGraph replays of the captured forward here - 702 vs 694 img/s on one stream, 985 vs 979 on four,
profiles/r04_batch1_graph.json: the ~110 launches of a batch-1 forward are bound by the kernels' own 8-13 us on the
device, not by the host's launch cost - and dropped the capture path again.)
    python scripts/bench_batch1.py [--dtype fp16p] [--n 96]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import synth  # noqa: E402
from dirtorch_amd import nets  # noqa: E402
from dirtorch_amd.test_dir import StreamPool  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='fp16p')
    ap.add_argument('--arch', default='resnet101')
    ap.add_argument('--n', type=int, default=96)
    args = ap.parse_args()
    net = nets.create_model(args.arch + '_rmac', pretrained='')
    net.load_state_dict(synth.synth_state_dict(args.arch, seed=7))
    net.compute_dtype = args.dtype
    net.cuda().eval()
    res = {'dtype': args.dtype, 'arch': args.arch}
    sizes = [tuple(int(v) for v in t.split('x')) for t in os.environ['B1_SIZES'].split(',')] if os.environ.get('B1_SIZES') else ((1024, 1024), (768, 1024))
    for H, W in sizes:
        g = torch.Generator(device='cuda').manual_seed(3)
        imgs = [torch.randint(0, 256, (1, H, W, 3), generator=g, dtype=torch.uint8, device='cuda') for _ in range(8)]
        ref = None
        for ns in ([int(v) for v in os.environ['B1_STREAMS'].split(',')] if os.environ.get('B1_STREAMS') else (1, 2, 3, 4, 6)):
            pool = StreamPool(ns)
            outs = []
            for rep in range(2):            # first pass: workspaces, lazy kernel attributes
                outs = []
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.n):
                    x = imgs[i % len(imgs)]
                    outs.append(pool.run(lambda: net(x), x))
                pool.join()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            got = torch.stack(outs[:8]).cpu()
            if ref is None:
                ref = got
            res['%dx%d_streams%d' % (H, W, ns)] = {'images_per_sec': round(args.n / dt, 1),
                                                  'max_abs_diff_vs_1_stream': float((got - ref).abs().max())}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
