#!/usr/bin/env python
"""(Test tooling, kept beside the tests because it drives the oracle: `python tests/precision_decomposition.py`.)
Where does the 16-bit error of the descriptors come from?  CPU only (the fp32 oracle with selected storage points
rounded), on the BatchNorm-calibrated ResNet-50 @ 224^2 case of tests/test_strict_gpu.py.  Storage points of the engine:
  W  the folded conv weights            A  the activations inside a bottleneck (t1, t2, the downsample branch)
  X  the trunk between bottlenecks (the residual carry, stem output included)
Prints 1 - cos of the descriptors vs the all-fp32 oracle for every combination that matters - the question being whether
an fp32 residual carry (W + A rounded, X kept in fp32) would bring a 16-bit path under the north-star 1e-4."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from oracle import dir_oracle as O  # noqa: E402


def features(sd, arch, x, dt, W, A, X):
    q = lambda t, on: t.to(dt).float() if on else t   # noqa: E731

    def conv(x, wkey, bn, stride, pad):
        w, b = O._fold(sd, wkey, bn, None)
        return F.conv2d(x, q(w, W), b, stride, pad)
    bottleneck, layers = O.ARCH[arch]
    x = q(x.float(), X)
    x = q(F.relu(conv(x, 'conv1.weight', 'bn1', 2, 3)), X)
    x = F.max_pool2d(x, 3, 2, 1)
    inplanes = 64
    for s, planes in enumerate((64, 128, 256, 512)):
        for j in range(layers[s]):
            pre = 'layer%d.%d' % (s + 1, j)
            stride = 2 if (j == 0 and s > 0) else 1
            res = x
            out = q(F.relu(conv(x, pre + '.conv1.weight', pre + '.bn1', 1, 0)), A)
            out = q(F.relu(conv(out, pre + '.conv2.weight', pre + '.bn2', stride, 1)), A)
            out = conv(out, pre + '.conv3.weight', pre + '.bn3', 1, 0)
            if j == 0:
                res = q(conv(x, pre + '.downsample.0.weight', pre + '.downsample.1', stride, 0), A)
            x = q(F.relu(out + res), X)
            inplanes = planes * 4
    return x


def main():
    arch, B, H, Wd, CB = 'resnet50', 8, 224, 224, 16
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    sd = O.calibrated_state_dict(arch, O.synth_images(99, CB, H, Wd), seed=7)
    x = O.synth_images(4, B, H, Wd)
    with torch.no_grad():
        rows = []
        for dname, dt in (('fp16', torch.float16), ('bf16', torch.bfloat16)):
            for label, (W, A, X) in (('W', (1, 0, 0)), ('A', (0, 1, 0)), ('X', (0, 0, 1)), ('W+A (fp32 residual carry)', (1, 1, 0)),
                                     ('A+X', (0, 1, 1)), ('W+A+X (the engine)', (1, 1, 1))):
                rows.append((dname, label, dt, W, A, X))
        f0 = features(sd, arch, x, torch.float16, 0, 0, 0)     # nothing rounded: the fp32 oracle
        # (GeM-pooled, L2-normalised trunk descriptor: the FC + L2 behind it is fp32 in every mode)
        head = lambda f: F.normalize(O.gem_pool(f, float(sd.get('adpool.p', torch.tensor(3.0)))).flatten(1), dim=1)   # noqa: E731
        d0 = head(f0)
        for dname, label, dt, W, A, X in rows:
            d = head(features(sd, arch, x, dt, W, A, X))
            print('%-5s rounded: %-28s 1 - cos (max over %d images) = %.2e' % (dname, label, B, float((1 - (d * d0).sum(1)).max())))


if __name__ == '__main__':
    main()
