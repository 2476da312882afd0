"""Deterministic synthetic checkpoints and images (no network: there are no real ones on the box).

Not part of the oracle: these are INPUT generators shared by tests/, tests/golden/make_golden.py,
bench.py and scripts/.  Everything is a pure function of (seed, key), so the golden script, the CPU
tests and the GPU box build bit-identical weights and images and the fixtures only have to hold the
reference's OUTPUTS.

    synth_state_dict        He-normal convs (reset_weights, dirtorch/nets/backbones/resnet.py:92-99),
                            non-trivial BatchNorm statistics, non-integer GeM exponent
    calibrated_state_dict   the same weights with BatchNorm running statistics CALIBRATED on a
                            synthetic image set (data-dependent init): every pre-activation is
                            zero-mean / unit-variance per channel over the calibration set, the way
                            a trained network's are, so descriptors of unrelated images are NOT
                            collinear (random-init descriptors have pairwise cosine 0.9996+,
                            SURVEY.md §7 "mAP parity is ill-conditioned with random weights")
    synth_images            normalised fp32 NCHW images with planted low-frequency structure
"""
import hashlib
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

ARCH = {
    'resnet18': (False, [2, 2, 2, 2]),
    'resnet50': (True, [3, 4, 6, 3]),
    'resnet101': (True, [3, 4, 23, 3]),
    'resnet152': (True, [3, 8, 36, 3]),
}
BN_EPS = 1e-5  # nn.BatchNorm2d default, resnet.py:117


def conv_specs(arch):
    """[(weight key, bn prefix, cout, cin, k, stride)] in the reference's state-dict order."""
    bottleneck, layers = ARCH[arch]
    exp = 4 if bottleneck else 1
    specs = [('conv1.weight', 'bn1', 64, 3, 7, 2)]
    inplanes = 64
    for s, planes in enumerate((64, 128, 256, 512)):
        for j in range(layers[s]):
            pre = 'layer%d.%d' % (s + 1, j)
            stride = 2 if (j == 0 and s > 0) else 1
            if bottleneck:
                specs += [(pre + '.conv1.weight', pre + '.bn1', planes, inplanes, 1, 1),
                          (pre + '.conv2.weight', pre + '.bn2', planes, planes, 3, stride),
                          (pre + '.conv3.weight', pre + '.bn3', planes * 4, planes, 1, 1)]
            else:
                specs += [(pre + '.conv1.weight', pre + '.bn1', planes, inplanes, 3, stride),
                          (pre + '.conv2.weight', pre + '.bn2', planes, planes, 3, 1)]
            if j == 0 and (stride != 1 or inplanes != planes * exp):
                specs.append((pre + '.downsample.0.weight', pre + '.downsample.1', planes * exp,
                              inplanes, 1, stride))
            inplanes = planes * exp
    return specs, inplanes


def _rng(seed, key):
    h = hashlib.sha256(('%d:%s' % (seed, key)).encode()).digest()
    return np.random.RandomState(int.from_bytes(h[:4], 'little'))


def synth_state_dict(arch, seed=0, out_dim=2048, gemp=2.7, pooling='gem', head='rmac'):
    """Deterministic per-key weights: identical wherever they are generated (golden script, tests,
    GPU box).  He-normal convs as reset_weights (resnet.py:92-99) but NON-trivial BatchNorm
    statistics, a non-integer GeM exponent, and a damped last BN per block so that activations
    stay O(1..100) through 33 residual blocks (fp16-safe)."""
    specs, feat = conv_specs(arch)
    sd = OrderedDict()
    for wkey, bn, cout, cin, k, _ in specs:
        n = k * k * cout
        sd[wkey] = torch.from_numpy(
            (_rng(seed, wkey).standard_normal((cout, cin, k, k)) * math.sqrt(2. / n)).astype(np.float32))
        r = _rng(seed, bn)
        last = bn.endswith('bn3') or (not ARCH[arch][0] and bn.endswith('bn2')) or 'downsample' in bn
        lo, hi = (0.25, 0.5) if last else (0.6, 1.2)
        sd[bn + '.weight'] = torch.from_numpy(r.uniform(lo, hi, cout).astype(np.float32))
        sd[bn + '.bias'] = torch.from_numpy((r.standard_normal(cout) * 0.1).astype(np.float32))
        sd[bn + '.running_mean'] = torch.from_numpy((r.standard_normal(cout) * 0.1).astype(np.float32))
        sd[bn + '.running_var'] = torch.from_numpy(r.uniform(0.6, 1.6, cout).astype(np.float32))
        sd[bn + '.num_batches_tracked'] = torch.tensor(1, dtype=torch.long)
    if head in ('fpn', 'fpn0'):   # rmac_resnet_fpn.py:24-46 (state-dict order of the module)
        dim1, dim2 = feat // 2, feat
        if head == 'fpn':
            for key, shape in (('conv1x5.weight', (dim1, dim2, 1, 1)), ('conv3c4.weight', (dim1, dim1, 3, 3))):
                n = shape[2] * shape[3] * shape[0]
                sd[key] = torch.from_numpy(
                    (_rng(seed, key).standard_normal(shape) * math.sqrt(2. / n)).astype(np.float32))
        sd['adpoolx5.p'] = torch.ones(1) * gemp
        sd['adpoolc4.p'] = torch.ones(1) * (gemp + 0.4)
        feat = dim1 + dim2
    elif head == 'rmac' and pooling.startswith('gem'):
        sd['adpool.p'] = torch.ones(1) * gemp
    r = _rng(seed, 'fc')
    bound = 1. / math.sqrt(feat)
    sd['fc.weight'] = torch.from_numpy(r.uniform(-bound, bound, (out_dim, feat)).astype(np.float32))
    sd['fc.bias'] = torch.from_numpy(r.uniform(-bound, bound, out_dim).astype(np.float32))
    return sd


def synth_images(seed, B, H, W):
    """Normalised fp32 NCHW images with planted low-frequency structure (not white noise)."""
    r = _rng(seed, 'img%dx%dx%d' % (B, H, W))
    yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing='ij')
    imgs = np.empty((B, 3, H, W), np.float32)
    for b in range(B):
        for c in range(3):
            f = r.uniform(1, 6, 2)
            ph = r.uniform(0, 6.28, 2)
            imgs[b, c] = (np.sin(f[0] * 6.28 * yy + ph[0]) * np.cos(f[1] * 6.28 * xx + ph[1])
                          + 0.35 * r.standard_normal((H, W)))
    return torch.from_numpy(imgs)


def calibrated_state_dict(arch, calib, seed=0, out_dim=2048, gemp=2.7):
    """synth_state_dict with every BatchNorm's running_mean / running_var replaced by the statistics
    of its input over the calibration batch `calib` [B,3,H,W] (one forward in fp32 on the CPU, the
    data-dependent initialisation a freshly built network gets from a few training-mode steps).
    The result is an ordinary reference-format state dict: the reference, the oracle and the engine
    all load it unchanged.  BatchNorm gammas keep the synthetic values, so residual branches stay
    damped and the trunk output O(1)."""
    sd = synth_state_dict(arch, seed=seed, out_dim=out_dim, gemp=gemp)
    bottleneck, layers = ARCH[arch]
    exp = 4 if bottleneck else 1

    def conv_bn(x, wkey, bn, stride, pad):
        y = F.conv2d(x, sd[wkey], None, stride, pad)
        mean = y.mean(dim=(0, 2, 3))
        var = y.var(dim=(0, 2, 3), unbiased=False)
        sd[bn + '.running_mean'] = mean.clone()
        sd[bn + '.running_var'] = var.clamp(min=1e-4).clone()
        return F.batch_norm(y, sd[bn + '.running_mean'], sd[bn + '.running_var'], sd[bn + '.weight'],
                            sd[bn + '.bias'], False, 0.0, BN_EPS)

    with torch.no_grad():
        x = F.relu(conv_bn(calib.float(), 'conv1.weight', 'bn1', 2, 3))
        x = F.max_pool2d(x, 3, 2, 1)
        inplanes = 64
        for s, planes in enumerate((64, 128, 256, 512)):
            for j in range(layers[s]):
                pre = 'layer%d.%d' % (s + 1, j)
                stride = 2 if (j == 0 and s > 0) else 1
                residual = x
                if bottleneck:
                    out = F.relu(conv_bn(x, pre + '.conv1.weight', pre + '.bn1', 1, 0))
                    out = F.relu(conv_bn(out, pre + '.conv2.weight', pre + '.bn2', stride, 1))
                    out = conv_bn(out, pre + '.conv3.weight', pre + '.bn3', 1, 0)
                else:
                    out = F.relu(conv_bn(x, pre + '.conv1.weight', pre + '.bn1', stride, 1))
                    out = conv_bn(out, pre + '.conv2.weight', pre + '.bn2', 1, 1)
                if j == 0 and (stride != 1 or inplanes != planes * exp):
                    residual = conv_bn(x, pre + '.downsample.0.weight', pre + '.downsample.1', stride, 0)
                x = F.relu(out + residual)
                inplanes = planes * exp
    return sd


def synth_descriptors(seed, n, D, clusters=6, noise=0.35):
    """Unit-norm fp32 rows with cluster structure (so that nearest neighbours are well separated and
    similarities take both signs): the inputs of the alpha-QE / DBA goldens."""
    r = _rng(seed, 'desc%dx%d' % (n, D))
    centers = r.standard_normal((clusters, D))
    x = centers[r.randint(0, clusters, n)] + 3.0 * noise * r.standard_normal((n, D))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)
