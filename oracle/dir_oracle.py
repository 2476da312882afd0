"""CPU oracle for the descriptor-extraction + ranking path of naver/deep-image-retrieval.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module, and only as the checker - the product (deep-image-retrieval_amd/) never does.

It restates, function by function, the arithmetic of the reference's hot path in plain fp32
PyTorch-CPU / NumPy, with no dependency on /root/reference at run time:

    resnet_features      dirtorch/nets/backbones/resnet.py:157-174 (ResNet.forward),
                         :67-87 (Bottleneck.forward), :29-44 (BasicBlock.forward), :134-141
    rmac_forward         dirtorch/nets/rmac_resnet.py:39-69 (ResNet_RMAC.forward)
    fpn_forward          dirtorch/nets/rmac_resnet_fpn.py:50-86 (ResNet_RMAC_FPN.forward)
    classifier_forward   dirtorch/nets/backbones/resnet.py:157-174 with fc_out > 0
    gem_pool             dirtorch/nets/layers/pooling.py:38-40
    resize_bilinear_u8   dirtorch/utils/transforms.py:133-185 (Scale -> PIL bilinear resize)
    pool                 dirtorch/utils/common.py:41-55
    whiten_features      dirtorch/utils/common.py:221-239
    matmul               dirtorch/utils/common.py:30-38
    expand_descriptors   dirtorch/test_dir.py:24-44 (alpha-QE / DBA)
    compute_average_precision   dirtorch/utils/evaluation.py:46-82
    eval_query_AP        dirtorch/datasets/generic.py:189-224 (+ get_relevants/get_junk :150-170)

Pinning: the reference ships no tests and no golden vectors, so the pins are outputs
of the reference itself, imported in the build container by tests/golden/make_golden.py and
committed under tests/golden/*.npz; tests/test_oracle_golden.py checks this module against them.
The third-party arithmetic underneath (PyTorch conv/BN/linear kernels, sklearn PCA attributes) is
"parity unpinned" beyond that: the reference pins no version or vector for it.

`quant=` emulates the engine's 16-bit storage points (weights after BN folding, every activation
tensor written to HBM) with fp32 accumulation, so kernel bugs can be told from precision drift.
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

# Input generators (deterministic synthetic checkpoints / images) live in tests/synth.py: they are not
# part of the checker.  Re-exported here because the tests address them as O.synth_*.
_TESTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests')
if _TESTS not in sys.path:
    sys.path.insert(0, _TESTS)
from synth import (ARCH, BN_EPS, conv_specs, synth_state_dict, synth_images,  # noqa: E402,F401
                   calibrated_state_dict)


# ---- 16-bit emulation ---------------------------------------------------------------------------
def _q(t, quant):
    if quant is None:
        return t
    if quant in ('fp16p', 'fp16pa') or isinstance(quant, tuple):   # outside the paired head DIR_FP16P is fp16
        quant = 'fp16'
    if quant == 'pair':   # the paired head of DIR_FP16P (csrc/conv_pair.hip): v ~ fp16(v) + fp16(v - fp16(v))
        hi = t.to(torch.float16).to(torch.float32)
        return hi + (t - hi).to(torch.float16).to(torch.float32)
    dt = {'bf16': torch.bfloat16, 'fp16': torch.float16}[quant]
    return t.to(dt).to(torch.float32)


def _fold(sd, wkey, bn, quant):
    """Conv weight/bias with eval-mode BatchNorm folded in (what the engine packs)."""
    w = sd[wkey].float()
    scale = sd[bn + '.weight'].float() / torch.sqrt(sd[bn + '.running_var'].float() + BN_EPS)
    bias = sd[bn + '.bias'].float() - sd[bn + '.running_mean'].float() * scale
    return _q(w * scale.view(-1, 1, 1, 1), quant), bias


def _conv_bn(sd, x, wkey, bn, stride, pad, quant):
    if quant is None:
        # the reference's own op order: conv, then BatchNorm2d in eval mode
        y = F.conv2d(x, sd[wkey].float(), None, stride, pad)
        return F.batch_norm(y, sd[bn + '.running_mean'].float(), sd[bn + '.running_var'].float(),
                            sd[bn + '.weight'].float(), sd[bn + '.bias'].float(), False, 0.0, BN_EPS)
    w, b = _fold(sd, wkey, bn, quant)
    return F.conv2d(x, w, b, stride, pad)


# ---- trunk ----------------------------------------------------------------------------------------
def resnet_features(sd, arch, x, quant=None, with_x4=False):
    """ResNet.forward up to layer4 (fc_out == 0 for *_rmac): [B,3,H,W] -> [B,C,h,w] fp32.
    with_x4 = the out_layer == -1 form (resnet.py:166-167): returns (layer3 map, layer4 map)."""
    bottleneck, layers = ARCH[arch]
    # quant = 'fp16p' (DIR_FP16P): the image, the stem (weights and output) and - inside the first `pair_stages` stages -
    # the weights of the 1x1 convs are fp16 PAIRS (~22 bits); everything else is 'fp16'.
    # quant = 'fp16pa' (DIRTORCH_AMD_PAIR_ACTS=1; what BasicBlock nets always get, their layer1 has no 1x1): there ALL
    # weights and the tensors between a block's convs (t1, t2, the downsample branch) are pairs as well.  Every block
    # OUTPUT (the 4P-wide residual carry) is a single fp16 plane in both.  ('fp16p' | 'fp16pa', pair_stages) moves the boundary.
    name = quant[0] if isinstance(quant, tuple) else quant
    fp16p = name in ('fp16p', 'fp16pa')
    acts = fp16p and (name == 'fp16pa' or not bottleneck)
    pair_stages = (quant[1] if isinstance(quant, tuple) else 1) if fp16p else 0
    tail_quant = 'fp16' if fp16p else quant
    quant = 'pair' if fp16p else quant
    x = _q(x.float(), quant)
    x = _q(F.relu(_conv_bn(sd, x, 'conv1.weight', 'bn1', 2, 3, quant)), quant)
    qa = quant if (not fp16p or acts) else 'fp16'     # tensors inside a block
    q3 = qa                                            # weights of the 3x3 convs
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    inplanes = 64
    exp = 4 if bottleneck else 1
    for s, planes in enumerate((64, 128, 256, 512)):
        for j in range(layers[s]):
            if fp16p and s >= pair_stages:
                quant = qa = q3 = tail_quant
            pre = 'layer%d.%d' % (s + 1, j)
            stride = 2 if (j == 0 and s > 0) else 1
            residual = x
            if bottleneck:
                out = _q(F.relu(_conv_bn(sd, x, pre + '.conv1.weight', pre + '.bn1', 1, 0, quant)), qa)
                out = _q(F.relu(_conv_bn(sd, out, pre + '.conv2.weight', pre + '.bn2', stride, 1, q3)), qa)
                out = _conv_bn(sd, out, pre + '.conv3.weight', pre + '.bn3', 1, 0, quant)
            else:
                out = _q(F.relu(_conv_bn(sd, x, pre + '.conv1.weight', pre + '.bn1', stride, 1, q3)), qa)
                out = _conv_bn(sd, out, pre + '.conv2.weight', pre + '.bn2', 1, 1, q3)
            if j == 0 and (stride != 1 or inplanes != planes * exp):
                residual = _q(_conv_bn(sd, x, pre + '.downsample.0.weight', pre + '.downsample.1',
                                       stride, 0, quant), qa)
            x = _q(F.relu(out + residual), tail_quant)     # block outputs: one plane in every mode
            inplanes = planes * exp
        if s == 2:
            x4 = x
    return (x4, x) if with_x4 else x


def gem_pool(x, p, eps=1e-6):
    """GeneralizedMeanPooling.forward with output_size 1: [B,C,h,w] -> [B,C,1,1]."""
    x = x.clamp(min=eps).pow(p)
    return F.adaptive_avg_pool2d(x, 1).pow(1. / p)


def center_bias_mask(b, size):
    bias = 1 + torch.tensor([[[[0, 0, 0, 0], [0, b, b, 0], [0, b, b, 0], [0, 0, 0, 0]]]], dtype=torch.float32)
    return F.interpolate(bias, size=size, mode='bilinear', align_corners=True)


def rmac_head(sd, feat, pooling='gem', norm_features=False, center_bias=0, without_fc=False):
    """rmac_resnet.py:52-68 from the trunk output on: returns [B,D], or [D] when B == 1."""
    x = feat
    if center_bias > 0:
        x = x * center_bias_mask(center_bias, x.shape[-2:])
    if pooling == 'max':
        x = F.adaptive_max_pool2d(x, 1)
    elif pooling == 'avg':
        x = F.adaptive_avg_pool2d(x, 1)
    elif pooling.startswith('gem'):
        x = gem_pool(x, sd['adpool.p'].float())
    else:
        raise ValueError(pooling)
    if norm_features:
        x = F.normalize(x, p=2, dim=1)
    x = x.squeeze()
    if x.dim() == 0:
        x = x.view(1)
    if not without_fc:
        x = F.linear(x, sd['fc.weight'].float(), sd['fc.bias'].float())
    return F.normalize(x, p=2, dim=-1)


def rmac_forward(sd, arch, x, pooling='gem', norm_features=False, center_bias=0, without_fc=False,
                 quant=None):
    with torch.no_grad():
        feat = resnet_features(sd, arch, x, quant)
        return rmac_head(sd, feat, pooling, norm_features, center_bias, without_fc)


def fpn_forward(sd, arch, x, mode=1, norm_features=False, without_fc=False, quant=None):
    """ResNet_RMAC_FPN.forward (rmac_resnet_fpn.py:50-86), eval mode, pooling 'gem':
    returns [B,D], or [D] when B == 1."""
    with torch.no_grad():
        x4, x5 = resnet_features(sd, arch, x, quant, with_x4=True)
        if mode == 1:
            c5 = F.interpolate(x5, size=x4.shape[-2:], mode='nearest')
            c5 = _q(F.relu(F.conv2d(c5, _q(sd['conv1x5.weight'].float(), quant))), quant)
            x4 = _q(x4 + c5, quant)
            x4 = _q(F.relu(F.conv2d(x4, _q(sd['conv3c4.weight'].float(), quant), padding=1)), quant)
        p5 = gem_pool(x5, sd['adpoolx5.p'].float())
        p4 = gem_pool(x4, sd['adpoolc4.p'].float())
        v = torch.cat((p4, p5), 1)
        if norm_features:
            v = F.normalize(v, p=2, dim=1)
        v = v.squeeze()
        if not without_fc:
            v = F.linear(v, sd['fc.weight'].float(), sd['fc.bias'].float())
        return F.normalize(v, p=2, dim=-1)


def classifier_forward(sd, arch, x, quant=None):
    """Plain ResNet.forward with fc_out > 0 (resnet.py:157-174): average pool + FC, [B, fc_out]."""
    with torch.no_grad():
        feat = resnet_features(sd, arch, x, quant)
        v = F.adaptive_avg_pool2d(feat, 1)
        v = v.view(v.size(0), -1)
        return F.linear(v, sd['fc.weight'].float(), sd['fc.bias'].float())


# ---- Scale (PIL bilinear resize of an 8-bit RGB image) ----------------------------------------------
# dirtorch/utils/transforms.py:133-185 calls img.resize(size, Image.BILINEAR).  The arithmetic lives
# in Pillow (unpinned by the reference; 12.2.0 here): src/libImaging/Resample.c, restated below -
# precompute_coeffs (triangle filter, support scaled by max(in/out, 1)), normalize_coeffs_8bpc
# (22-bit fixed point), then a horizontal and a vertical pass that each round to uint8.
PIL_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size, out_size):
    """(xmin[out], count[out], kk[out, ksize] int32) for one axis, box = the whole image."""
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = 0.0 + (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    ss = 1.0 / filterscale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    k = np.zeros((out_size, ksize), dtype=np.float64)
    ww = np.zeros(out_size, dtype=np.float64)
    for x in range(ksize):
        arg = np.abs((x + xmin - center + 0.5) * ss)
        w = np.where(arg < 1.0, 1.0 - arg, 0.0)
        w = np.where(x < xmax, w, 0.0)
        k[:, x] = w
        ww = ww + w
    nz = ww != 0.0
    k[nz] = k[nz] / ww[nz, None]
    kk = np.trunc(np.where(k < 0, -0.5, 0.5) + k * float(1 << PIL_PRECISION_BITS)).astype(np.int32)
    return xmin.astype(np.int32), xmax.astype(np.int32), kk


def _pil_resample_axis(img, out_size, axis):
    """One 8-bit pass of ImagingResampleHorizontal/Vertical_8bpc along `axis` of an HWC array."""
    xmin, cnt, kk = pil_bilinear_coeffs(img.shape[axis], out_size)
    a = np.moveaxis(img, axis, 0).astype(np.int64)
    acc = np.full((out_size,) + a.shape[1:], 1 << (PIL_PRECISION_BITS - 1), dtype=np.int64)
    for x in range(kk.shape[1]):
        idx = np.minimum(xmin + x, a.shape[0] - 1)          # taps past the count carry weight 0
        wgt = np.where(x < cnt, kk[:, x], 0).astype(np.int64)
        acc += a[idx] * wgt.reshape((-1,) + (1,) * (a.ndim - 1))
    out = np.clip(acc >> PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, ow, oh):
    """PIL.Image.resize((ow, oh), Image.BILINEAR) of a uint8 HWC image; a pass whose size does not
    change is skipped, as ImagingResample does."""
    img = np.ascontiguousarray(img)
    if ow != img.shape[1]:
        img = _pil_resample_axis(img, ow, 1)
    if oh != img.shape[0]:
        img = _pil_resample_axis(img, oh, 0)
    return img


# ---- post-processing --------------------------------------------------------------------------------
def pool(x, pooling='mean', gemp=3):
    """common.pool: list of [N,D] tensors -> [N,D] (identity for a single scale)."""
    if len(x) == 1:
        return x[0]
    x = torch.stack(x, dim=0)
    if pooling == 'mean':
        return torch.mean(x, dim=0)
    elif pooling == 'gem':
        def sympow(x, p, eps=1e-6):
            s = torch.sign(x)
            return (x * s).clamp(min=eps).pow(p) * s
        x = sympow(x, gemp)
        x = torch.mean(x, dim=0)
        return sympow(x, 1 / gemp)
    raise ValueError("Bad pooling mode: " + str(pooling))


class PCAParams(object):
    """The four attributes common.transform reads from an sklearn PCA (common.py:224-228)."""

    def __init__(self, mean_, components_, explained_variance_, whiten=True):
        self.mean_ = mean_
        self.components_ = components_
        self.explained_variance_ = explained_variance_
        self.whiten = whiten


def fit_pca(X, whiten=True):
    """Exact PCA by SVD on centred data (what sklearn's full solver computes, up to sign)."""
    X = np.asarray(X, dtype=np.float64)
    mean = X.mean(axis=0)
    U, S, Vt = np.linalg.svd(X - mean, full_matrices=False)
    var = (S ** 2) / (X.shape[0] - 1)
    return PCAParams(mean.astype(np.float32), Vt.astype(np.float32), var.astype(np.float32), whiten)


def whiten_features(X, pca, l2norm=True, whitenp=0.5, whitenv=None, whitenm=1.0):
    """common.transform + whiten_features (use_sklearn=True branch)."""
    if pca.mean_ is not None:
        X = X - pca.mean_
    Xt = np.dot(X, pca.components_[:whitenv].T)
    if pca.whiten:
        Xt = Xt / (whitenm * np.power(pca.explained_variance_[:whitenv], whitenp))
    if l2norm:
        Xt = Xt / np.expand_dims(np.linalg.norm(Xt, axis=1), axis=1)
    return Xt


def matmul(A, B):
    """Q x N similarity scores as fp32 NumPy (common.matmul)."""
    return np.dot(np.asarray(A), np.asarray(B).T)


def expand_descriptors(descs, db=None, alpha=0, k=0):
    """alpha query expansion / database augmentation (test_dir.py:24-44): mean of a descriptor and its
    k most similar rows of `db` (weights sim**alpha), L2-normalised; db=None expands the set against
    itself with the diagonal of the similarity zeroed."""
    assert k >= 0 and alpha >= 0
    if k == 0:
        return descs
    descs = np.asarray(descs)
    n = descs.shape[0]
    db_descs = np.asarray(db) if db is not None else descs
    sim = matmul(descs, db_descs)
    if db is None:
        sim[np.diag_indices(n)] = 0
    idx = np.argpartition(sim, int(-k), axis=1)[:, int(-k):]
    out = np.zeros_like(descs)
    for i in range(n):
        rows = np.vstack([descs[i]] + [db_descs[j, :] * sim[i, j] ** alpha for j in idx[i]])
        new_q = np.mean(rows, axis=0)
        out[i] = new_q / np.linalg.norm(new_q)
    return out


# ---- ranking / AP (revisited Oxford/Paris protocol) ---------------------------------------------------
def compute_average_precision(positive_ranks):
    """Trapezoidal AP over sorted zero-based ranks of the positives (evaluation.py:46-82)."""
    ap = 0.0
    n = len(positive_ranks)
    if not n:
        return ap
    step = 1.0 / n
    for i, rank in enumerate(positive_ranks):
        left = 1.0 if not rank else i / rank
        right = (i + 1) / (rank + 1)
        ap += (left + right) * step / 2
    return ap


def eval_query_AP(scores, easy, hard, junk):
    """{'easy','medium','hard'} AP of one query from its score row (generic.py:209-224).
    easy/hard/junk are index lists as in the revisitop gnd pickle; ties rank by descending index
    (np.argsort(...)[::-1])."""
    N = scores.shape[0]
    d = {}
    for mode in ('easy', 'medium', 'hard'):
        if mode == 'easy':
            rel, jk = list(easy), list(junk) + list(hard)
        elif mode == 'medium':
            rel, jk = list(easy) + list(hard), list(junk)
        else:
            rel, jk = list(hard), list(junk) + list(easy)
        gt = -np.ones(N, dtype=np.int8)
        gt[rel] = 1
        gt[jk] = 0
        keep = gt != 0
        if np.sum(gt[keep] > 0) == 0:
            d[mode] = -1
        else:
            gt2, s2 = gt[keep], scores[keep]
            gt_sorted = gt2[np.argsort(s2)[::-1]]
            d[mode] = compute_average_precision(np.where(gt_sorted == 1)[0])
    return d


def mean_ap(scores, gnd):
    """mAP-easy/medium/hard over all queries; gnd = list of {'easy','hard','junk'} dicts
    (the aggregation of dirtorch/test_dir.py:153-167: queries with AP -1 are skipped)."""
    aps = [eval_query_AP(scores[q], g['easy'], g['hard'], g['junk']) for q, g in enumerate(gnd)]
    res = {}
    for mode in ('easy', 'medium', 'hard'):
        v = [a[mode] for a in aps if a[mode] >= 0]
        res['mAP-' + mode] = float(np.mean(v)) if v else float('nan')
    return res


def cosine(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1, np.asarray(a).shape[-1])
    b = np.asarray(b, dtype=np.float64).reshape(-1, np.asarray(b).shape[-1])
    return np.sum(a * b, axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
