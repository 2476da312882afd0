#!/usr/bin/env python
"""PCIe-inclusive extraction rate: images start in pinned HOST memory (raw uint8 NHWC, what the
loader hands over), are copied to the GPU on a side stream into one of two device buffers while the
previous batch runs dir_forward on the main stream.  Not bench.py's `value` (that one starts with
inputs resident in HBM); reported in DESIGN.md §5 next to it."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--arch', default='resnet101')
    ap.add_argument('--fmt', default='u8', choices=['u8', 'f32'])
    args = ap.parse_args()
    import synth
    from dirtorch_amd import nets
    net = nets.create_model(args.arch + '_rmac', pretrained='')
    net.load_state_dict(synth.synth_state_dict(args.arch, seed=7))
    net.cuda().eval()
    B, S = args.batch, args.size
    if args.fmt == 'u8':
        host = [torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
    else:   # the reference's feed: normalised fp32 NCHW (common.py:214)
        host = [torch.randn(B, 3, S, S).pin_memory() for _ in range(2)]
    dev = [torch.empty_like(h, device='cuda') for h in host]
    net(dev[0].copy_(host[0]))
    net.autotune = False
    copy_stream = torch.cuda.Stream()
    main_stream = torch.cuda.current_stream()
    ready = [torch.cuda.Event() for _ in range(2)]     # H2D of buffer i finished
    freed = [torch.cuda.Event() for _ in range(2)]     # forward over buffer i finished
    out = torch.empty(args.steps * B, net.out_dim, device='cuda')

    def run(K):
        for i in range(2):
            freed[i].record(main_stream)
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(freed[0])
            dev[0].copy_(host[0], non_blocking=True)
            ready[0].record(copy_stream)
        for k in range(K):
            cur, nxt = k % 2, (k + 1) % 2
            if k + 1 < K:
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(freed[nxt])
                    dev[nxt].copy_(host[nxt], non_blocking=True)
                    ready[nxt].record(copy_stream)
            main_stream.wait_event(ready[cur])
            out[k * B:(k + 1) * B] = net(dev[cur])
            freed[cur].record(main_stream)
        torch.cuda.synchronize()

    run(3)
    t0 = time.perf_counter()
    run(args.steps)
    el = time.perf_counter() - t0
    bytes_per_img = host[0].numel() * host[0].element_size() / B
    print(json.dumps({'images_per_s_pcie_inclusive': round(args.steps * B / el, 1), 'feed': args.fmt,
                      'h2d_GBps': round(args.steps * B * bytes_per_img / el / 1e9, 2),
                      'batch': B, 'size': S, 'steps': args.steps}))


if __name__ == '__main__':
    main()
