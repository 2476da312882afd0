#!/usr/bin/env python
"""Debug: per-scale distance of the engine (fp16p) from the CPU oracle on one 1200^2 picture, both feeds, both stems."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle'), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
import synth
import dir_oracle as O
import bench
from dirtorch_amd import nets, ops
from dirtorch_amd.utils import common, transforms

torch.set_num_threads(bench.cpu_allotted())
arch, S = 'resnet101', int(os.environ.get('EXP_S', 1200))
sd = synth.calibrated_state_dict(arch, synth.synth_images(99, 1, S, S), seed=7)
pic = bench.to_uint8_nhwc(synth.synth_images(4, 1, S, S))
scales = [transforms.Scale(0.7071), None, transforms.Scale(1.4142)]
sizes = [(S, S) if sc is None else sc.target_size((S, S)) for sc in scales]
net = nets.create_model(arch + '_rmac', pretrained='')
net.load_state_dict(sd)
net.compute_dtype = os.environ.get('EXP_DTYPE', 'fp16p')
net.cuda().eval()
g = torch.Generator(device='cuda').manual_seed(99)
B = int(os.environ.get('EXP_B', 4))
img = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device='cuda', generator=g)
img[0] = pic[0].cuda()
per_e, per_o, per_f = [], [], []
for (w_, h_) in sizes:
    x = img if (w_, h_) == (S, S) else ops.resize_bilinear_u8(img, (w_, h_))
    d8 = net(x).reshape(B, -1)[:1].cpu()
    u = pic[0].numpy() if (w_, h_) == (S, S) else O.resize_bilinear_u8(pic[0].numpy(), w_, h_)
    print('resize identical:', bool(np.array_equal(x[0].cpu().numpy(), u)), x.shape)
    xf = bench.normalise_uint8(torch.from_numpy(u)[None])
    df = net(xf.cuda()).reshape(1, -1).cpu()
    do = O.rmac_forward(sd, arch, xf).reshape(1, -1)
    print('scale %dx%d: 1-cos  u8 feed %.3e   fp32 feed %.3e   u8 vs fp32 feed %.3e   overflow %s' % (
        w_, h_, 1 - O.cosine(d8.numpy(), do.numpy())[0], 1 - O.cosine(df.numpy(), do.numpy())[0], 1 - O.cosine(d8.numpy(), df.numpy())[0], net.overflowed()))
    per_e.append(d8.cuda()); per_o.append(do); per_f.append(df.cuda())
pe = common.l2_normalize(common.pool(per_e, 'gem', 3)).cpu().numpy()
pf = common.l2_normalize(common.pool(per_f, 'gem', 3)).cpu().numpy()
po = torch.nn.functional.normalize(O.pool(per_o, 'gem', 3), dim=1).numpy()
print('pooled: u8 feed %.3e  fp32 feed %.3e' % (1 - O.cosine(pe, po)[0], 1 - O.cosine(pf, po)[0]))
