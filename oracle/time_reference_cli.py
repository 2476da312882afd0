#!/usr/bin/env python
"""Time the REFERENCE's own CLI on the CPU (build container only: the GPU box has no /root/reference).

TEST/MEASUREMENT INFRASTRUCTURE, like the rest of oracle/: nothing in the product imports this.

    python oracle/time_reference_cli.py [--arch resnet50] [--size 224] [--n 64] [--threads-torch N]

Runs, unmodified, `python -m dirtorch.extract_features --dataset 'ImageList("list.txt")' --checkpoint
synth.pt --output out.npy --gpu -1` (dirtorch/extract_features.py:82-124 -> :26-68 ->
test_dir.extract_image_features, test_dir.py:47-94: batch 1, 8 loader workers) on N synthetic PNGs and
a synthetic checkpoint (tests/synth.py), BASELINE.json configs[0], and a bare `net(x)` loop over the same
images to separate decode / loader time from compute.  Two things the reference needs in this container
 are supplied from OUTSIDE its source tree, which is never modified or copied:

  * torchvision is not installed: a 4-name stand-in (Compose, ToTensor, Normalize, Lambda - the only
    names dirtorch/utils/transforms.py uses on this path, :33, :536-551, :617-623) is injected into
    sys.modules; ToTensor = u8 HWC -> f32 CHW / 255, Normalize = (x - mean[c]) / std[c];
  * torch >= 2.6 refuses to unpickle a checkpoint holding non-tensor objects under weights_only=True
    (dirtorch/utils/common.py:121 passes no flag): torch.load is wrapped to default to
    weights_only=False for this trusted, locally written file.

Prints one JSON line (core count, torch threads, images/s of the CLI and of the bare forward loop) and,
with --check, the cosine between the CLI's .npy and this repo's oracle on the same files.
"""
import argparse
import functools
import json
import os
import runpy
import sys
import tempfile
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'


def install_torchvision_shim():
    class Compose(object):
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, img):
            for t in self.transforms:
                img = t(img)
            return img

    class ToTensor(object):
        def __call__(self, pic):
            a = np.asarray(pic)
            if a.ndim == 2:
                a = a[:, :, None]
            return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)

    class Normalize(object):
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, t):
            mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
            std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
            return (t - mean) / std

    class Lambda(object):
        def __init__(self, fn):
            self.fn = fn

        def __call__(self, img):
            return self.fn(img)

    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    tvt.Compose, tvt.ToTensor, tvt.Normalize, tvt.Lambda = Compose, ToTensor, Normalize, Lambda
    tv.transforms = tvt
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.transforms'] = tvt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='resnet50')
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--n', type=int, default=64)
    ap.add_argument('--threads-torch', type=int, default=0, help='torch.set_num_threads (0 = all visible CPUs)')
    ap.add_argument('--check', action='store_true')
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit('this script needs /root/reference (build container only)')
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import synth
    from PIL import Image

    ncpu = len(os.sched_getaffinity(0))
    torch.set_num_threads(args.threads_torch or ncpu)
    tmp = tempfile.mkdtemp(prefix='refcli_')
    os.environ.setdefault('DB_ROOT', tmp)
    r = np.random.RandomState(0)
    names = []
    for i in range(args.n):
        Image.fromarray(r.randint(0, 256, (args.size, args.size, 3)).astype(np.uint8)).save(
            os.path.join(tmp, 'im%03d.png' % i))
        names.append('im%03d.png' % i)
    open(os.path.join(tmp, 'list.txt'), 'w').write('\n'.join(names) + '\n')
    sd = synth.synth_state_dict(args.arch, seed=7, gemp=3.0)
    ck = os.path.join(tmp, 'synth.pt')
    torch.save({'model_options': dict(arch=args.arch + '_rmac', out_dim=2048, pooling='gem', gemp=3),
                'state_dict': sd}, ck)

    install_torchvision_shim()
    torch.load = functools.partial(torch.load, weights_only=False)
    sys.path.insert(0, REF)
    out = os.path.join(tmp, 'out.npy')
    argv = ['dirtorch.extract_features', '--dataset', 'ImageList("%s", root="%s")' % (os.path.join(tmp, 'list.txt'), tmp),
            '--checkpoint', ck, '--output', out, '--gpu', '-1']
    old = sys.argv
    sys.argv = argv
    t0 = time.perf_counter()
    runpy.run_module('dirtorch.extract_features', run_name='__main__')
    cli_s = time.perf_counter() - t0
    sys.argv = old
    feats = np.load(out)
    assert feats.shape == (args.n, 2048), feats.shape

    # bare forward loop (compute only), the same network object the CLI builds, batch 1
    import dirtorch.nets as ref_nets
    net = ref_nets.create_model(args.arch + '_rmac', pretrained='', out_dim=2048, pooling='gem', gemp=3)
    net.load_state_dict(sd)
    net.eval()
    x = torch.randn(1, 3, args.size, args.size)
    with torch.no_grad():
        net(x)
        t0 = time.perf_counter()
        for _ in range(args.n):
            net(x)
        fwd_s = time.perf_counter() - t0
    res = {'reference_cli': 'python -m dirtorch.extract_features --gpu -1', 'arch': args.arch, 'size': args.size,
           'images': args.n, 'cpus_visible': ncpu, 'torch_threads': torch.get_num_threads(),
           'cli_seconds': round(cli_s, 2), 'cli_images_per_s': round(args.n / cli_s, 2),
           'forward_only_images_per_s': round(args.n / fwd_s, 2), 'torch': torch.__version__}
    if args.check:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import dir_oracle as O
        mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
        rows = []
        for n in names[:8]:
            img = torch.from_numpy(np.asarray(Image.open(os.path.join(tmp, n)).convert('RGB')).copy())
            rows.append(O.rmac_forward(sd, args.arch, ((img.permute(2, 0, 1).float() / 255 - mean) / std)[None]).reshape(1, -1))
        res['oracle_vs_cli_min_cosine'] = float(O.cosine(torch.cat(rows).numpy(), feats[:8]).min())
    print(json.dumps(res))


if __name__ == '__main__':
    main()
