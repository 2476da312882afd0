"""The DIR_FP16P stem on the RAW uint8 feed (csrc/stem_u8.hip): ToTensor + Normalize (dirtorch/utils/transforms.py:617-623)
folded into conv1 + bn1 + ReLU + MaxPool (dirtorch/nets/backbones/resnet.py:115-119,158-161).

The reference normalises first and zero-pads AFTER (the conv's padding=3 acts on the normalised tensor), so the fold needs
a border-class bias table; these tests hold the kernel to the reference's arithmetic in fp64 - interior and border pixels
separately - through the C ABI (dir_stem_pool_u8), and the engine on the uint8 feed to the engine on the fp32 feed and
to the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def join(p):
    return p[0].float() + p[1].float()


def reference_stem(u8, w, scale, bias, mean=MEAN, std=STD):
    """fp64: ToTensor (/255), Normalize, conv 7x7 s2 p3 (zero padding in NORMALISED space), folded BN, ReLU, max-pool 3x3 s2 p1
    -> [B, PH, PW, 64]."""
    x = (u8.double() / 255.0 - torch.tensor(mean).double()) / torch.tensor(std).double()       # [B,H,W,3]
    x = x.permute(0, 3, 1, 2)
    y = F.conv2d(x, w.double(), None, 2, 3) * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1)
    return F.max_pool2d(F.relu(y), 3, 2, 1).permute(0, 2, 3, 1)


def make_case(seed, B, H, W, flat=False):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    if flat:      # a constant image: every interior output equal, every border class visible as its own value
        u8[:] = torch.tensor([200, 90, 30], dtype=torch.uint8)
    w = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    scale = 0.5 + torch.rand(64, generator=g)
    bias = 0.3 * torch.randn(64, generator=g)
    return u8, w, scale, bias


# (even widths take the RAW form - the stem reads the image bytes itself -, odd ones the prep_input_u8 plane; odd heights both)
SIZES = [(2, 64, 96), (1, 75, 61), (1, 224, 224), (1, 7, 7), (2, 8, 9), (1, 10, 33), (3, 300, 130), (1, 513, 767), (2, 61, 76), (1, 7, 8), (2, 513, 640)]


@pytest.mark.parametrize('B,H,W', SIZES, ids=['%dx%dx%d' % s for s in SIZES])
def test_stem_pool_u8_vs_fp64_reference(B, H, W, monkeypatch):
    from dirtorch_amd import ops
    u8, w, scale, bias = make_case(31, B, H, W)
    ref = reference_stem(u8, w, scale, bias)
    got = join(ops.stem_pool_u8(u8.cuda(), w, scale, bias, MEAN, STD)).cpu().double()
    assert got.shape == ref.shape
    tol = 2e-6 * max(1.0, float(ref.abs().max()))
    err = (got - ref).abs()
    # border pooled pixels (their windows hold conv outputs whose 7x7 windows leave the image) and the interior, apart
    inner = err[:, 2:-2, 2:-2] if min(err.shape[1:3]) > 4 else err[:, :0, :0]
    print('\n[stem-u8] %dx%dx%d: max |d| %.3e (interior %.3e) of max |ref| %.3e' % (
        B, H, W, float(err.max()), float(inner.max()) if inner.numel() else 0.0, float(ref.abs().max())))
    assert float(err.max()) < tol, (float(err.max()), tol)
    # segment length does not change a bit: 1 = independent 3 x 15 tiles, 2 / 3 = short walks, default = 8 tiles per segment
    base = ops.stem_pool_u8(u8.cuda(), w, scale, bias, MEAN, STD)
    for seg in (1, 2, 3):
        alt = ops.stem_pool_u8(u8.cuda(), w, scale, bias, MEAN, STD, seg_tiles=seg)
        assert torch.equal(alt[0], base[0]) and torch.equal(alt[1], base[1]), seg
    # ... nor where the patch comes from: the two-kernel form (prep_input_u8 writes the u / 256 plane, the stem reads it by LDS-DMA)
    # against the default for an even W, where the stem converts the image bytes itself
    monkeypatch.setenv('DIRTORCH_AMD_STEM_U8_PREP', '1')
    alt = ops.stem_pool_u8(u8.cuda(), w, scale, bias, MEAN, STD)
    assert torch.equal(alt[0], base[0]) and torch.equal(alt[1], base[1]), 'prep'
    monkeypatch.delenv('DIRTORCH_AMD_STEM_U8_PREP')
    # ... nor does the workgroup shape: one 8-wave workgroup per CU on 8 x 32 conv tiles (the default: two 4-wave ones on 4 x 32)
    monkeypatch.setenv('DIRTORCH_AMD_STEM_U8_WG8', '1')
    for seg in (0, 1):
        alt = ops.stem_pool_u8(u8.cuda(), w, scale, bias, MEAN, STD, seg_tiles=seg)
        assert torch.equal(alt[0], base[0]) and torch.equal(alt[1], base[1]), ('wg8', seg)


def test_border_classes_on_a_constant_image():
    """A constant image makes the missing - mean / std of every padded tap visible: without the border table the first two and
    the last conv rows / columns are wrong by O(1).  The result must equal the fp64 reference everywhere, and the folded form
    must NOT equal a naive fold that ignores the border (so the test can tell the table is in use)."""
    from dirtorch_amd import ops
    for H, W in ((40, 52), (41, 53)):       # even / odd: different last-row classes
        u8, w, scale, bias = make_case(5, 1, H, W, flat=True)
        ref = reference_stem(u8, w, scale, bias)
        got = join(ops.stem_pool_u8(u8.cuda(), w, scale, bias, MEAN, STD)).cpu().double()
        assert float((got - ref).abs().max()) < 2e-6 * max(1.0, float(ref.abs().max()))
        # naive fold: normalise the zero padding too (= pad the raw image with 0 and apply the affine map everywhere)
        x = (u8.double() / 255.0).permute(0, 3, 1, 2)
        x = F.pad(x, (3, 3, 3, 3))
        x = (x - torch.tensor(MEAN).double().view(1, 3, 1, 1)) / torch.tensor(STD).double().view(1, 3, 1, 1)
        y = F.conv2d(x, w.double(), None, 2, 0) * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1)
        naive = F.max_pool2d(F.relu(y), 3, 2, 1).permute(0, 2, 3, 1)
        assert float((naive - ref).abs().max()) > 1e-2


def test_stem_pool_u8_is_more_exact_than_one_fp16_plane():
    """The filter is a PAIR: with its lo plane dropped (11-bit weights) the result is off at the 1e-4 level."""
    from dirtorch_amd import ops
    u8, w, scale, bias = make_case(9, 1, 96, 96)
    ref = reference_stem(u8, w, scale, bias)
    got = join(ops.stem_pool_u8(u8.cuda(), w, scale, bias, MEAN, STD)).cpu().double()
    wq = (w * scale.view(-1, 1, 1, 1) * 256.0 / (255.0 * torch.tensor(STD).view(1, 3, 1, 1))).half().float()
    wq = wq / (scale.view(-1, 1, 1, 1) * 256.0 / (255.0 * torch.tensor(STD).view(1, 3, 1, 1)))
    one_plane = reference_stem(u8, wq, scale, bias)
    assert float((got - ref).abs().max()) < 0.05 * float((one_plane - ref).abs().max())


@pytest.mark.parametrize('arch,B,H,W', [('resnet50', 3, 224, 224), ('resnet18', 2, 97, 131), ('resnet101', 1, 512, 640)],
                         ids=['r50_224', 'r18_97x131', 'r101_512x640'])
def test_engine_uint8_feed_fp16p_vs_fp32_feed_and_oracle(arch, B, H, W):
    """dir_forward(DIR_IMG_U8_NHWC) in fp16p runs prep_input_u8 + stem_pool_u8; its descriptors must agree with the same
    engine fed the reference's normalised fp32 tensor (the generic paired stem: image pair, three MFMAs per term) and with
    the fp32 CPU oracle on that tensor (rmac_resnet.py:39-69)."""
    import os
    import dir_oracle as O
    from dirtorch_amd import _lib, nets
    sd = O.calibrated_state_dict(arch, O.synth_images(99, 4, min(H, 160), min(W, 160)), seed=7)
    net = nets.create_model(arch + '_rmac', pretrained='')
    net.load_state_dict(sd)
    net.compute_dtype = 'fp16p'
    net.cuda().eval()
    g = torch.Generator().manual_seed(17)
    u8 = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    mean, std = torch.tensor(net.rgb_means), torch.tensor(net.rgb_stds)
    xf = ((u8.float() / 255.0 - mean) / std).permute(0, 3, 1, 2).contiguous()
    net.set_profiling(True)
    a = net(u8.cuda()).cpu().numpy().reshape(B, -1)
    used = {r['name']: r['kernel'] for r in net.get_profile()}
    net.set_profiling(False)
    # (an even width: the stem reads the image itself, there is no prep launch; an odd one: prep_input_u8 + the stem)
    assert used.get('conv1+maxpool') == 'stem_pool_u8' and used.get('prep_input') == (None if W % 2 == 0 else 'prep_input_u8'), used
    b = net(xf.cuda()).cpu().numpy().reshape(B, -1)
    ref = O.rmac_forward(sd, arch, xf).numpy().reshape(B, -1)
    e_ab, e_a, e_b = 1 - O.cosine(a, b), 1 - O.cosine(a, ref), 1 - O.cosine(b, ref)
    print('\n[stem-u8 engine] %s %dx%d: u8 feed vs fp32 feed %.3e | vs oracle: u8 feed %.3e, fp32 feed %.3e' % (
        arch, H, W, e_ab.max(), e_a.max(), e_b.max()))
    assert np.isfinite(a).all() and not net.overflowed()
    assert np.all(e_a < 1e-4) and np.all(e_ab < 5e-5)
    # the A/B switch restores the generic paired stem on the uint8 feed
    os.environ['DIRTORCH_AMD_NO_STEM_U8'] = '1'
    _lib.reload_env()
    try:
        net2 = nets.create_model(arch + '_rmac', pretrained='')
        net2.load_state_dict(sd)
        net2.compute_dtype = 'fp16p'
        net2.cuda().eval()
        net2.set_profiling(True)
        c = net2(u8.cuda()).cpu().numpy().reshape(B, -1)
        used2 = {r['name']: r['kernel'] for r in net2.get_profile()}
    finally:
        del os.environ['DIRTORCH_AMD_NO_STEM_U8']
        _lib.reload_env()
    assert used2.get('conv1+maxpool') == 'stem_pool_pair', used2
    assert np.all(1 - O.cosine(a, c) < 5e-5)


@pytest.mark.parametrize('H,W', [(96, 128), (97, 130), (75, 61)], ids=['96x128', '97x130', '75x61_odd_w'])
def test_fp32_feed_stem_reads_the_image_itself(H, W, monkeypatch):
    """The generic paired stem on the fp32 NCHW feed (net(x) on the reference's normalised tensor, rmac_resnet.py:39): with an even
    width the kernel splits the image into its (hi, lo) planes on its own - no prep_input_pair launch - and the descriptors equal the
    two-kernel form's bit for bit (DIRTORCH_AMD_STEM_U8_PREP=1), which an odd width still takes."""
    import dir_oracle as O
    from dirtorch_amd import nets
    sd = O.synth_state_dict('resnet50', seed=7)
    x = O.synth_images(5, 3, H, W).cuda()

    def run():
        net = nets.create_model('resnet50_rmac', pretrained='')
        net.load_state_dict(sd)
        net.compute_dtype = 'fp16p'
        net.cuda().eval()
        net.set_profiling(True)
        d = net(x).clone()
        return d, {r['name']: r['kernel'] for r in net.get_profile()}
    d_raw, used = run()
    assert used.get('conv1+maxpool') == 'stem_pool_pair'
    assert ('prep_input' in used) == (W % 2 == 1), used
    monkeypatch.setenv('DIRTORCH_AMD_STEM_U8_PREP', '1')
    d_prep, used2 = run()
    assert used2.get('prep_input') == 'prep_input_pair'
    assert torch.equal(d_raw, d_prep)
    ref = O.rmac_forward(sd, 'resnet50', x.cpu()).numpy()
    assert np.all(1 - O.cosine(d_raw.cpu().numpy(), ref) < 1e-5)
