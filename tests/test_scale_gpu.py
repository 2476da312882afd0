"""Parity at the BASELINE sizes, engine vs the CPU oracle.

The kernels that decide the bench number (conv_wreg / conv_persist / the 256x256 tiles) are only
selected at network scale, so the golden cases (<= 128 px) never reach them.  Here the oracle runs
the same images on the CPU (fp32; ~1 s per 1024^2 image) and the engine must agree:

  * descriptors: 1 - cos < 1e-4 against the fp32 oracle (the north-star gate) on the synthetic
    checkpoint of the golden tests, and - on a BatchNorm-calibrated checkpoint whose descriptors are
    not collinear - within a small factor of what an IDEAL 16-bit-storage implementation loses (the
    oracle's quant= emulation of the engine's storage points);
  * trunk feature map: per 64-pixel tile (the granularity of the persistent kernels), so that a
    half-landed first tile cannot hide inside a whole-map norm.

Measured values are printed (pytest -s) and recorded in BASELINE.md.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_net(arch, sd, dtype):
    from dirtorch_amd import nets
    net = nets.create_model(arch + '_rmac', pretrained='')
    net.load_state_dict(sd)
    net.compute_dtype = dtype
    net.cuda()
    return net.eval()


_CACHE = {}


def cached(key, make):
    """The CPU oracle is most of this file's wall time and the fp32 reference of a case does not depend on the
    engine's dtype: compute it once per (case) and share it between the bf16 and fp16 runs."""
    if key not in _CACHE:
        _CACHE[key] = make()
    return _CACHE[key]


def oracle_desc(sd, arch, x, quant=None, chunk=8):
    import dir_oracle as O
    rows = [O.rmac_forward(sd, arch, x[i:i + chunk], quant=quant).reshape(-1, sd['fc.weight'].shape[0])
            for i in range(0, x.shape[0], chunk)]
    return torch.cat(rows).numpy()


# (tag, arch, B, H, W): config B (headline), its odd-size variant, config A, config E scales
SIZES = [
    ('r101_1024_b2', 'resnet101', 2, 1024, 1024),
    ('r101_1023x767_b1', 'resnet101', 1, 1023, 767),
    ('r50_224_b64', 'resnet50', 64, 224, 224),
    ('r101_1200_b1', 'resnet101', 1, 1200, 1200),
    ('r101_1697_b1', 'resnet101', 1, 1697, 1697),
]


@pytest.mark.parametrize('dtype', ['bf16', 'fp16', 'fp16p'])
@pytest.mark.parametrize('tag,arch,B,H,W', SIZES, ids=[s[0] for s in SIZES])
def test_descriptor_vs_oracle_at_baseline_sizes(tag, arch, B, H, W, dtype):
    import dir_oracle as O
    sd = O.synth_state_dict(arch, seed=7)
    x = O.synth_images(4, B, H, W)
    net = make_net(arch, sd, dtype)
    with torch.no_grad():
        got = net(x.cuda()).cpu().numpy().reshape(B, -1)
    ref = cached(('desc', tag), lambda: oracle_desc(sd, arch, x))
    assert np.isfinite(got).all()
    err = 1 - O.cosine(got, ref)
    print('\n[scale] %s %s: 1-cos vs fp32 oracle max %.3e mean %.3e' % (tag, dtype, err.max(), err.mean()))
    assert np.all(err < 1e-4), err.max()


# the odd-size variant of config B and the two outer scales of config E (the middle scales are test_pair_gpu.py's)
LITERAL_SIZES = [('r101_1023x767', 'resnet101', 2, 1023, 767), ('r101_1200', 'resnet101', 1, 1200, 1200),
                 ('r101_1697', 'resnet101', 1, 1697, 1697)]


@pytest.mark.parametrize('tag,arch,B,H,W', LITERAL_SIZES, ids=[s[0] for s in LITERAL_SIZES])
def test_fp16p_meets_the_stated_tolerance_on_the_calibrated_checkpoint(tag, arch, B, H, W):
    """The north-star tolerance LITERALLY - 1 - cos < 1e-4 against the fp32 CPU oracle, no derived allowance - for the
    host's and bench.py's default dtype (fp16p: fp16 with the paired head) on the BatchNorm-calibrated checkpoint, the
    conditioned network on which 16-bit storage error is visible (the He-init checkpoint's descriptors are collinear and
    flatter every format), at the sizes of BASELINE configs[1] (odd-size variant) and configs[4] (1200^2, 1697^2)."""
    import dir_oracle as O
    sd = cached(('calib-sd', arch, H, W), lambda: O.calibrated_state_dict(arch, O.synth_images(99, B, H, W), seed=7))
    x = O.synth_images(4, B, H, W)
    net = make_net(arch, sd, 'fp16p')
    with torch.no_grad():
        got = net(x.cuda()).cpu().numpy().reshape(B, -1)
    ref = cached(('calib-ref', arch, H, W), lambda: oracle_desc(sd, arch, x, chunk=1))
    assert np.isfinite(got).all() and not net.overflowed()
    err = 1 - O.cosine(got, ref)
    print('\n[scale-literal] %s fp16p, calibrated checkpoint: 1-cos vs fp32 oracle max %.3e' % (tag, err.max()))
    assert np.all(err < 1e-4), err.max()


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('arch,B,H,W,CB', [('resnet101', 2, 1024, 1024, 2), ('resnet50', 16, 224, 224, 16)],
                         ids=['r101_1024', 'r50_224'])
def test_calibrated_checkpoint_error_vs_ideal_16bit(arch, B, H, W, CB, dtype):
    """On a checkpoint whose BatchNorm statistics are calibrated (descriptors of different images are
    not collinear, so 1 - cos is not flattered by a shared mean direction) the engine may lose what
    16-bit activation storage loses - measured by the oracle's emulation of the same storage points -
    and no more: engine error <= 3 x emulation error, and the engine sits as close to the emulation as
    the emulation sits to fp32."""
    import dir_oracle as O
    sd = cached(('calib-sd', arch, H, W), lambda: O.calibrated_state_dict(arch, O.synth_images(99, CB, H, W), seed=7))
    x = O.synth_images(4, B, H, W)
    net = make_net(arch, sd, dtype)
    with torch.no_grad():
        got = net(x.cuda()).cpu().numpy().reshape(B, -1)
    ref = cached(('calib-ref', arch, H, W), lambda: oracle_desc(sd, arch, x))
    emu = oracle_desc(sd, arch, x, quant=dtype)
    e_got = 1 - O.cosine(got, ref)
    e_emu = 1 - O.cosine(emu, ref)
    e_ge = 1 - O.cosine(got, emu)
    pair = float(np.max((ref @ ref.T) - np.eye(B)))
    print('\n[scale-calibrated] %s %dx%d %s: engine-vs-fp32 %.3e, ideal-16bit-vs-fp32 %.3e, engine-vs-ideal %.3e, '
          'max cosine between different images %.4f' % (arch, H, W, dtype, e_got.max(), e_emu.max(), e_ge.max(), pair))
    assert np.isfinite(got).all()
    assert e_got.max() <= 3 * e_emu.max() + 1e-6, (e_got.max(), e_emu.max())
    assert e_ge.max() <= 4 * e_emu.max() + 1e-6, (e_ge.max(), e_emu.max())   # independent roundings add: ~2x


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_trunk_tiles_at_headline_size(dtype):
    """forward_features of ResNet-101 @ 1024^2 (B = 2) against the oracle with the engine's storage
    points emulated, judged PER 64-PIXEL TILE of the NHWC map (pixel index m = (b, h, w), the unit a
    persistent workgroup of conv_wreg / conv_persist produces): relative L2 and max-abs of every tile.
    One corrupted first tile in any of the 101 layers lands in a few such tiles as an O(1) error, while
    legitimate 16-bit rounding drift stays at the whole-map level."""
    import dir_oracle as O
    arch, B, S = 'resnet101', 2, 1024
    sd = O.synth_state_dict(arch, seed=7)
    x = O.synth_images(4, B, S, S)
    net = make_net(arch, sd, dtype)
    feat = net.forward_features(x.cuda()).float().cpu()              # [B, 32, 32, 2048] NHWC
    with torch.no_grad():
        ref = O.resnet_features(sd, arch, x, quant=dtype).permute(0, 2, 3, 1).contiguous()
    assert feat.shape == ref.shape == (B, 32, 32, 2048)
    C = feat.shape[-1]
    g = feat.reshape(-1, 64, C)
    r = ref.reshape(-1, 64, C)
    tile_rel = ((g - r).flatten(1).norm(dim=1) / r.flatten(1).norm(dim=1)).numpy()
    tile_max = (g - r).flatten(1).abs().max(dim=1).values.numpy()
    whole = float((feat - ref).norm() / ref.norm())
    scale = float(ref.abs().max())
    print('\n[scale-tiles] %s: whole-map rel L2 %.3e; per-tile rel L2 max %.3e median %.3e; per-tile max-abs %.3e '
          '(map max %.3e)' % (dtype, whole, tile_rel.max(), np.median(tile_rel), tile_max.max(), scale))
    # whole map: one 16-bit rounding per stored activation compounded over 101 layers (measured on
    # MI355X: 8.5e-3 bf16, 1.07e-3 fp16; the 17-50 layer nets of test_model_gpu.py sit at 4e-3 / 5e-4)
    tol = 1.2e-2 if dtype == 'bf16' else 1.5e-3
    assert whole < tol, whole
    # the sharp criterion: rounding drift is UNIFORM over the map (measured max / median = 1.01), a tile
    # made of garbage has rel L2 ~ 1
    assert tile_rel.max() < 1.5 * np.median(tile_rel), (tile_rel.max(), np.median(tile_rel))
    assert tile_max.max() < 0.05 * scale, (tile_max.max(), scale)


# ---- the configuration bench.py times: kernel mix asserted, rows and tiles against the CPU oracle ------------
# dir_conv_heuristic (engine.hip run_conv) picks DIFFERENT kernels at batch 32 than at batch 2: the 512x128
# LDS-patch 3x3 (conv2 of layers 2-4, the dominant kernel of the bench), the register-stationary conv3, the
# persistent 256x256 conv1 / conv3 (+ its three-deep form for layer4's conv1), the fused layer2 seam and the
# two-source GEMMs.  Descriptors do not depend on the batch an image travels in, so three rows of the batch
# (first, middle, last: the first / an interior / the last tile of every persistent workgroup's walk) are
# enough, and the CPU oracle only has to run three images.
TIMED_MIX = {
    32: {'layer2.1.conv2': '512x128_patch3x3w', 'layer3.7.conv2': '512x128_patch3x3w', 'layer4.1.conv2': '512x128_patch3x3w',
         'layer2.3.conv3': '64x512_wreg1x1', 'layer3.9.conv3': '64x512_wreg1x1',
         'layer3.5.conv1': '256x256_persist1x1_x3', 'layer4.2.conv3': '256x256_persist1x1',
         'layer4.1.conv1': '256x256_persist1x1_x3', 'layer2.0.conv2': '256x128_patchs2', 'layer3.0.conv2': '256x256_w4x4'},
    16: {'layer2.1.conv2': '512x128_patch3x3w', 'layer3.7.conv2': '512x128_patch3x3w',
         'layer2.3.conv3': '64x512_wreg1x1', 'layer3.9.conv3': '64x512_wreg1x1',
         'layer3.5.conv1': '256x256_persist1x1_x3', 'layer2.0.conv2': '256x128_patchs2'},
}
ROWS = {32: (0, 13, 31), 16: (0, 13)}     # batch 16 = the first 16 images of the batch-32 case: rows 0 and 13 reuse its oracle results


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('B', [32, 16])
def test_timed_configuration_vs_oracle(B, dtype):
    """ResNet-101 @ 1024^2 at the batch bench.py runs (32; 16 has another layer4 mix): (1) the profile shows
    the kernels the bench number is made of, (2) descriptors of rows {first, middle, last} agree with the fp32
    oracle to the north-star 1e-4 cosine, (3) their trunk maps agree with the storage-emulating oracle per
    64-pixel tile (reference: dirtorch/nets/rmac_resnet.py:39-69, backbones/resnet.py:67-87,157-174)."""
    import dir_oracle as O
    arch, S = 'resnet101', 1024
    rows = list(ROWS[B])
    sd = O.synth_state_dict(arch, seed=7)
    x = cached(('timed-x', 32), lambda: O.synth_images(4, 32, S, S))[:B]
    net = make_net(arch, sd, dtype)
    xg = x.cuda()
    net.set_profiling(True)
    with torch.no_grad():
        got = net(xg).cpu().numpy()
    used = {r['name']: r['kernel'] for r in net.get_profile()}
    net.set_profiling(False)
    for layer, variant in TIMED_MIX[B].items():
        assert used.get(layer) == 'conv_igemm<%s>' % variant, (layer, used.get(layer))
    assert used.get('layer2.1.c3c1') == 'conv_c3c1<128>' and used.get('layer1.0.ds+c3c1') == 'conv_c3c1<64,ds>', used
    for s in (2, 3, 4):
        assert used.get('layer%d.0.ds+conv3' % s, '').endswith('/dual>'), used
    assert np.isfinite(got).all()
    ref = cached(('timed-desc', 32), lambda: oracle_desc(sd, arch, cached(('timed-x', 32), None)[list(ROWS[32])], chunk=1))[:len(rows)]
    err = 1 - O.cosine(got[rows], ref)
    print('\n[timed] B=%d %s: 1-cos vs fp32 oracle rows %s: %s' % (B, dtype, rows, err))
    assert np.all(err < 1e-4), err
    # batch independence of the rows NOT sent to the oracle: every row of the batch equals the row computed
    # in the oracle-checked batch-2 configuration to 16-bit-rounding noise (other kernels, same arithmetic)
    with torch.no_grad():
        small = torch.cat([net(xg[i:i + 2]).reshape(2, -1) for i in range(0, B, 2)]).cpu().numpy()
    assert np.all(1 - O.cosine(got, small) < 2e-5), (1 - O.cosine(got, small)).max()
    del small
    if B != 32:
        return
    # trunk maps of the three rows, per 64-pixel tile (the bench batch only: the CPU side costs ~5 s per image)
    feat = net.forward_features(xg)[rows].float().cpu()              # [3, 32, 32, 2048]
    with torch.no_grad():
        fref = torch.cat([O.resnet_features(sd, arch, x[i:i + 1], quant=dtype) for i in rows]).permute(0, 2, 3, 1).contiguous()
    g = feat.reshape(-1, 64, feat.shape[-1])
    r = fref.reshape(-1, 64, feat.shape[-1])
    tile_rel = ((g - r).flatten(1).norm(dim=1) / r.flatten(1).norm(dim=1)).numpy()
    whole = float((feat - fref).norm() / fref.norm())
    print('[timed] B=%d %s: trunk whole-map rel L2 %.3e, per-tile max %.3e median %.3e' % (B, dtype, whole, tile_rel.max(), np.median(tile_rel)))
    assert whole < (1.2e-2 if dtype == 'bf16' else 1.5e-3), whole
    assert tile_rel.max() < 1.5 * np.median(tile_rel), (tile_rel.max(), np.median(tile_rel))


# ---- the same gate for the HEADLINE format: fp16p, batch 32, on the conditioned checkpoint, literally ---------------
# (round-5 review, "What's weak" 1a: the kernels the bench number is made of - the paired stem, conv_pair's conv1 of block 0, the
# paired-weight seams of layer1, in-place identity blocks in layers 3-4 - met the oracle only inside bench.py's own parity leg)
FP16P_HEAD = {
    'f32': {'conv1+maxpool': 'stem_pool_pair'},     # (round 6: no prep launch either - the paired stem splits the fp32 image itself)
    'u8': {'conv1+maxpool': 'stem_pool_u8'},        # (no prep launch: at an even width the stem converts the image bytes itself)
}
FP16P_MIX = {'layer1.0.conv1': 'conv_pair<128x64_xw>', 'layer1.0.ds+c3c1': 'conv_c3c1<64,ds,wp>',
             'layer1.1.c3c1': 'conv_c3c1<64,wp>', 'layer1.2.c3c1': 'conv_c3c1<64,wp>', 'layer2.1.c3c1': 'conv_c3c1<128>'}


@pytest.mark.parametrize('feed', ['f32', 'u8'])
def test_timed_configuration_fp16p_vs_oracle_on_the_calibrated_checkpoint(feed):
    """bench.py's default line: ResNet-101 @ 1024^2, batch 32, fp16p, on BOTH feeds (the reference's normalised fp32 NCHW
    tensor, rmac_resnet.py:39; the raw uint8 NHWC image with ToTensor / Normalize on the device, transforms.py:617-623 - the
    feed the drop-in CLIs use).  (1) the profile shows the timed kernel mix, (2) rows {0, 13, 31} meet 1 - cos < 1e-4 against
    the fp32 CPU oracle on the BatchNorm-CALIBRATED checkpoint - the north-star tolerance, no derived allowance."""
    import os
    import dir_oracle as O
    arch, S, B = 'resnet101', 1024, 32
    rows = list(ROWS[B])
    assert not os.environ.get('DIRTORCH_AMD_NO_INPLACE'), 'the timed configuration writes identity blocks in place'
    sd = cached(('calib-sd', arch, S, S), lambda: O.calibrated_state_dict(arch, O.synth_images(99, 2, S, S), seed=7))
    net = make_net(arch, sd, 'fp16p')
    x = cached(('timed-x', 32), lambda: O.synth_images(4, 32, S, S))
    if feed == 'u8':
        mean, std = torch.tensor(net.rgb_means).view(1, 3, 1, 1), torch.tensor(net.rgb_stds).view(1, 3, 1, 1)
        u8 = ((x * std + mean) * 255.0).round().clamp(0, 255).to(torch.uint8)          # [B, 3, S, S]
        xin = u8.permute(0, 2, 3, 1).contiguous().cuda()                                 # raw NHWC, as PIL hands it over
        # what the reference feeds its network for these pixels: ToTensor (/ 255) then Normalize, in fp32
        xo = ((u8[rows].float() / 255.0) - mean) / std
    else:
        xin = x.cuda()
        xo = x[rows]
    net.set_profiling(True)
    with torch.no_grad():
        got = net(xin).cpu().numpy()
    used = {r['name']: r['kernel'] for r in net.get_profile()}
    net.set_profiling(False)
    for layer, variant in TIMED_MIX[B].items():
        assert used.get(layer) == 'conv_igemm<%s>' % variant, (layer, used.get(layer))
    for layer, kern in list(FP16P_MIX.items()) + list(FP16P_HEAD[feed].items()):
        assert used.get(layer) == kern, (layer, used.get(layer))
    for s_ in (2, 3, 4):
        assert used.get('layer%d.0.ds+conv3' % s_, '').endswith('/dual>'), used
    assert 'prep_input' not in used, used
    assert np.isfinite(got).all() and not net.overflowed()
    ref = cached(('timed-calib-desc', feed), lambda: oracle_desc(sd, arch, xo, chunk=1))
    err = 1 - O.cosine(got[rows], ref)
    print('\n[timed-fp16p] B=%d fp16p, %s feed, calibrated checkpoint: 1-cos vs fp32 oracle rows %s: %s' % (B, feed, rows, err))
    assert np.all(err < 1e-4), err
