"""Block- and model-level parity on the MI355X.

  * engine descriptors vs the REFERENCE's own outputs (tests/golden/model_goldens.npz):
    cosine >= 1 - 1e-4 (the north-star tolerance), same shapes incl. the (D,) squeeze at B == 1;
  * engine trunk features vs the oracle with the engine's 16-bit storage points emulated
    (tight: only summation order differs);
  * a BASELINE-size forward (ResNet-101 @ 1024x1024) checked through size-independent properties.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_oracle_golden import CASES, case_inputs  # noqa: E402


def make_net(arch, opts, sd, dtype):
    from dirtorch_amd import nets
    net = nets.create_model(arch + '_rmac', pretrained='', **opts)
    net.load_state_dict(sd)
    net.compute_dtype = dtype
    net.cuda()
    return net.eval()


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_descriptor_vs_reference_golden(case, dtype, model_goldens):
    import dir_oracle as O
    tag, arch, opts, gemp, B, H, W = case
    sd, x = case_inputs(*case)
    net = make_net(arch, opts, sd, dtype)
    with torch.no_grad():
        desc = net(x.cuda())
    torch.cuda.synchronize()
    gold = model_goldens[tag + '.desc']
    got = desc.cpu().numpy()
    assert got.shape == gold.shape, (got.shape, gold.shape)     # (D,) when B == 1
    assert np.isfinite(got).all()
    np.testing.assert_allclose(np.linalg.norm(got.reshape(-1, got.shape[-1]), axis=1), 1.0, atol=1e-5)
    cos = O.cosine(got, gold)
    assert np.all(1 - cos < 1e-4), '%s %s: 1-cos = %s' % (tag, dtype, 1 - cos)


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('arch,H,W', [('resnet18', 64, 80), ('resnet50', 97, 75)])
def test_trunk_features_vs_quantised_oracle(arch, H, W, dtype):
    import dir_oracle as O
    sd = O.synth_state_dict(arch, seed=7)
    x = O.synth_images(11, 2, H, W)
    net = make_net(arch, {}, sd, dtype)
    feat = net.forward_features(x.cuda()).float().cpu().permute(0, 3, 1, 2)
    with torch.no_grad():
        ref = O.resnet_features(sd, arch, x, quant=dtype)
        ref32 = O.resnet_features(sd, arch, x)
    assert feat.shape == ref.shape
    rel = float((feat - ref).norm() / ref.norm())
    rel32 = float((feat - ref32).norm() / ref32.norm())
    # relative L2 error of the whole feature map; one 16-bit rounding per stored activation
    # (bf16 2^-9, fp16 2^-12 per element) compounded over 17-50 layers.  Measured on MI355X:
    # 2.3e-3..3.8e-3 (bf16), 3.1e-4..4.9e-4 (fp16); 1-ulp flips decorrelate the engine from the
    # emulation as much as from the fp32 reference, so both get the same bound.
    tol = 8e-3 if dtype == 'bf16' else 1e-3
    assert rel < tol and rel32 < tol, (rel, rel32)


def test_uint8_input_path_matches_float_path():
    import dir_oracle as O
    sd = O.synth_state_dict('resnet18', seed=7)
    net = make_net('resnet18', {}, sd, 'fp16')
    g = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (2, 70, 90, 3), generator=g, dtype=torch.uint8)
    mean, std = torch.tensor(net.rgb_means), torch.tensor(net.rgb_stds)
    xf = ((u8.float() / 255.0 - mean) / std).permute(0, 3, 1, 2).contiguous()
    a = net(u8.cuda()).cpu().numpy()
    b = net(xf.cuda()).cpu().numpy()
    assert np.all(1 - O.cosine(a, b) < 1e-6)


def test_batch_composition_is_irrelevant():
    """Descriptor of an image does not depend on its batch neighbours (image-parallel sharding).
    Bit-exact while the same kernels run; layers with few output tiles switch to split-K at small
    batch (another fp32 association of the same products), which moves a descriptor by ~1e-7."""
    import dir_oracle as O
    sd = O.synth_state_dict('resnet50', seed=7)
    net = make_net('resnet50', {}, sd, 'bf16')
    x = O.synth_images(3, 3, 96, 64).cuda()
    full = net(x).cpu()
    again = net(x).cpu()
    assert torch.equal(full, again)                      # run-to-run: bit-identical (fixed slice order)
    for i in range(3):
        one = net(x[i:i + 1]).cpu()
        assert one.shape == (2048,)
        assert torch.equal(one, net(x[i:i + 1]).cpu())
        assert float(1 - torch.dot(one, full[i])) < 1e-6, i
        assert float((one - full[i]).abs().max()) < 2e-4, i


@pytest.mark.parametrize('dtype', ['fp16', 'bf16', 'fp16p'])     # (fp16p on this uint8 feed: stem_u8.hip's hand-written waits under overlap)
def test_forwards_overlapping_on_streams_are_bit_identical(dtype):
    """The batch-1 extraction loop issues forwards round-robin on a few HIP streams (test_dir.StreamPool: one 1024^2
    image cannot fill 256 CUs).  Every trunk map must equal the single-stream one bit for bit - the same kernels run on
    the same data, only their interleaving changes.  (Round 3 found a kernel whose emitted code read an LDS-DMA stage
    ahead of its hand-off barrier once per ~2 400 launches under such overlap - csrc/dir_common.h ring_barrier,
    scripts/exp_stream_race*.py, tests/test_isa_audit.py.  This is the in-suite guard; the scripts have the statistics.)"""
    import dir_oracle as O
    from dirtorch_amd.test_dir import StreamPool
    sd = O.synth_state_dict('resnet50', seed=7)
    net = make_net('resnet50', {}, sd, dtype)
    g = torch.Generator(device='cuda').manual_seed(3)
    imgs = [torch.randint(0, 256, (1, 512, 640, 3), generator=g, dtype=torch.uint8, device='cuda') for _ in range(4)]
    refs = [net.forward_features(x).clone() for x in imgs]
    dref = [net(x).clone() for x in imgs]
    torch.cuda.synchronize()
    pool = StreamPool(4)
    assert len(pool.streams) == 4
    for rep in range(3):
        outs = [pool.run(lambda x=imgs[i % 4]: net.forward_features(x), imgs[i % 4]) for i in range(64)]
        descs = [pool.run(lambda x=imgs[i % 4]: net(x), imgs[i % 4]) for i in range(16)]
        pool.join()
        torch.cuda.synchronize()
        bad = [i for i, o in enumerate(outs) if not torch.equal(o, refs[i % 4])]
        assert not bad, 'forwards %s of repetition %d differ from the single-stream trunk map' % (bad[:8], rep)
        assert all(torch.equal(d, dref[i % 4]) for i, d in enumerate(descs))


def test_workspace_and_argument_errors():
    import ctypes
    from dirtorch_amd import _lib
    import dir_oracle as O
    sd = O.synth_state_dict('resnet18', seed=7)
    net = make_net('resnet18', {}, sd, 'bf16')
    x = O.synth_images(3, 1, 64, 64).cuda()
    net(x)
    out = torch.empty(1, 2048, device='cuda')
    small = torch.empty(1024, dtype=torch.uint8, device='cuda')
    with pytest.raises(_lib.DirError) as ei:
        _lib.call('dir_forward', net._engine, _lib.ptr(x), 1, 64, 64, 0, _lib.ptr(out), _lib.ptr(small),
                  small.numel(), _lib.stream_ptr())
    assert ei.value.code == -4 and 'workspace' in str(ei.value)
    with pytest.raises(_lib.DirError):
        net(torch.zeros(1, 3, 4, 4, device='cuda'))       # smaller than the 7x7 stem
    with pytest.raises(ValueError):
        net(torch.zeros(1, 4, 64, 64, device='cuda'))
    # missing tensor is reported by name
    bad = {k: v for k, v in sd.items() if k != 'layer2.0.bn1.running_var'}
    net2 = make_net('resnet18', {}, sd, 'bf16')
    net2.load_state_dict(bad, strict=False)
    net2._state.pop('layer2.0.bn1.running_var')
    e = ctypes.c_void_p()
    with pytest.raises(_lib.DirError) as ei:
        net2._engine = None
        net2(x)
    assert 'layer2.0.bn1' in str(ei.value)


def test_baseline_size_forward_properties():
    """ResNet-101 @ 1024x1024 (BASELINE config B) through properties that need no CPU oracle run:
    unit norm, finiteness, batch-independence, determinism, and agreement of the autotuned
    tile choice with the heuristic one (different tilings, same arithmetic up to summation order)."""
    import dir_oracle as O
    sd = O.synth_state_dict('resnet101', seed=7)
    net = make_net('resnet101', {}, sd, 'bf16')
    x = O.synth_images(4, 2, 1024, 1024).cuda()
    d1 = net(x).cpu()
    d2 = net(x).cpu()
    assert d1.shape == (2, 2048) and torch.isfinite(d1).all()
    assert torch.equal(d1, d2)
    np.testing.assert_allclose(d1.norm(dim=1).numpy(), 1.0, atol=1e-5)
    single = net(x[1:2]).cpu()
    assert float(1 - (single * d1[1]).sum()) < 1e-6
    net.autotune = True
    d3 = net(x).cpu()
    assert np.all(1 - O.cosine(d3.numpy(), d1.numpy()) < 1e-5)
    # non-square, odd size (1023 x 767): the real workload is variable H x W
    y = O.synth_images(5, 1, 1023, 767).cuda()
    net.autotune = False
    dy = net(y).cpu()
    assert dy.shape == (2048,) and torch.isfinite(dy).all()


def test_batches_beyond_the_32bit_descriptor_limit_are_split():
    """An activation tensor of one engine call must stay below 2^31 bytes; the host object splits
    larger batches and the result equals the per-chunk calls bit for bit."""
    import dir_oracle as O
    sd = O.synth_state_dict('resnet50', seed=7)
    net = make_net('resnet50', {}, sd, 'bf16')
    H = W = 1024
    limit = net.max_batch(H, W)
    assert limit == 63                       # layer1 maps: 256 x 256 x 256 x 2 B per image
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randint(0, 256, (limit + 3, H, W, 3), generator=g, dtype=torch.uint8, device='cuda')
    full = net(x)
    assert full.shape == (limit + 3, 2048)
    assert torch.equal(full[:limit], net(x[:limit]))
    assert torch.equal(full[limit:], net(x[limit:]))
    from dirtorch_amd import _lib
    with pytest.raises(_lib.DirError) as ei:      # the raw engine call refuses, loudly
        ws = net._workspace(limit + 3, H, W)
        out = torch.empty(limit + 3, 2048, device='cuda')
        _lib.call('dir_forward', net._engine, _lib.ptr(x), limit + 3, H, W, _lib.DIR_IMG_U8_NHWC, _lib.ptr(out),
                  _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    assert '2^31' in str(ei.value)


def test_register_stationary_conv3_inside_the_network(monkeypatch):
    """At batch 8 x 1024^2 the heuristic routes every layer2 / layer3 conv3 of ResNet-50 to the
    persistent register-stationary kernel (conv_wreg.hip); the same network with those layers pinned to
    a tiled variant through a tuning table must give the same feature map up to 16-bit rounding."""
    import dir_oracle as O
    monkeypatch.setenv('DIRTORCH_AMD_C3C1', '0')      # (the fused seam kernel would take layer2's conv3 instead)
    sd = O.synth_state_dict('resnet50', seed=7)
    B = 8
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randint(0, 256, (B, 1024, 1024, 3), generator=g, dtype=torch.uint8, device='cuda')

    def features(table):
        net = make_net('resnet50', {}, sd, 'bf16')
        net._build_engine()
        if table:
            net.import_tuning(table)
        f = net.forward_features(x).float()
        prof_names = None
        if not table:
            net.set_profiling(True)
            net(x)
            prof_names = {r['name']: r['kernel'] for r in net.get_profile()}
        return f, prof_names

    got, used = features('')
    assert used['layer3.2.conv3'] == 'conv_igemm<64x512_wreg1x1>' and used['layer2.1.conv3'] == 'conv_igemm<64x512_wreg1x1>'
    M2, M3 = B * 128 * 128, B * 64 * 64
    table = ''.join('layer2.%d.conv3 %d 128x256_w2x4_s3_k32\n' % (i, M2) for i in range(4)) + \
        ''.join('layer3.%d.conv3 %d 128x256_w2x4_s3_k32\n' % (i, M3) for i in range(6))
    ref, _ = features(table)
    assert torch.isfinite(got).all()
    rel = float((got - ref).norm() / ref.norm())
    # bf16 1-ulp flips downstream of a different fp32 summation order: the same bound as the trunk test
    # against the quantised oracle (measured 4e-3)
    assert rel < 8e-3, rel
    assert float((got - ref).abs().max()) < 0.05 * float(ref.abs().max())


def test_fused_seams_inside_the_network(monkeypatch):
    """ResNet-50 at 8 x 1024^2 with and without the fused kernels of the bottleneck tails:
      * conv_c3c1.hip: conv3 (+ residual + ReLU) -> next block's conv1 in layer1 / layer2, the DS form for
        layer1's first block (downsample folded in as extra K);
      * conv_igemm.hip DUAL: conv3 + downsample of the first block of layers 2-4 as one two-source GEMM.
    The feature map must equal the un-fused network's up to 16-bit rounding flips (the fused kernels feed
    conv1 the same rounded tensor they store; the folded downsample skips one rounding of the residual),
    and the profile must show the fused launches in place of the ones they replace."""
    import dir_oracle as O
    sd = O.synth_state_dict('resnet50', seed=7)
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randint(0, 256, (8, 1024, 1024, 3), generator=g, dtype=torch.uint8, device='cuda')

    def run(fused, dtype):
        if fused:
            monkeypatch.delenv('DIRTORCH_AMD_C3C1', raising=False)
            monkeypatch.delenv('DIRTORCH_AMD_NO_DUAL', raising=False)
        else:
            monkeypatch.setenv('DIRTORCH_AMD_C3C1', '0')
            monkeypatch.setenv('DIRTORCH_AMD_NO_DUAL', '1')
        net = make_net('resnet50', {}, sd, dtype)
        f = net.forward_features(x).float()
        net.set_profiling(True)
        d = net(x)
        names = {r['name']: r['kernel'] for r in net.get_profile()}
        return f, d, names

    for dtype, tol in (('bf16', 8e-3), ('fp16', 1e-3)):
        f_ref, d_ref, n_ref = run(False, dtype)
        f_fus, d_fus, n_fus = run(True, dtype)
        assert not any('c3c1' in k or 'dual' in k for k in n_ref.values())
        assert 'layer1.0.downsample' in n_ref and 'layer2.0.downsample' in n_ref
        assert n_fus.get('layer1.0.ds+c3c1') == 'conv_c3c1<64,ds>' and n_fus.get('layer1.1.c3c1') == 'conv_c3c1<64>', n_fus
        assert n_fus.get('layer2.1.c3c1') == 'conv_c3c1<128>', n_fus
        for s in (2, 3, 4):
            # (layer2: K = 128 + 256 fits a wave's registers - conv_wregd.hip, round 6; layers 3-4: conv_persist.hip's DUAL ring)
            assert n_fus.get('layer%d.0.ds+conv3' % s) == ('conv_igemm<64x256_wregd1x1/dual>' if s == 2 else
                                                           'conv_igemm<256x256_persist1x1_x3/dual>'), n_fus
            assert 'layer%d.0.downsample' % s not in n_fus and 'layer%d.0.conv3' % s not in n_fus
        assert 'layer1.1.conv1' not in n_fus and 'layer1.0.conv3' not in n_fus and 'layer1.0.downsample' not in n_fus
        # the layer1 -> layer2 boundary is a seam too (conv1 of layer2.0 is 1x1 stride 1); layer2.0 then closes
        # with the two-source GEMM, so layer2.1's conv1 runs on its own, as do the wider stage boundaries
        assert n_fus.get('layer1.2.c3c1') == 'conv_c3c1<64>' and 'layer2.0.conv1' not in n_fus and 'layer1.2.conv3' not in n_fus
        assert 'layer2.1.conv1' in n_fus and 'layer3.0.conv1' in n_fus and 'layer2.3.conv3' in n_fus
        assert torch.isfinite(f_fus).all()
        rel = float((f_fus - f_ref).norm() / f_ref.norm())
        assert rel < tol, (dtype, rel)
        assert np.all(1 - O.cosine(d_fus.cpu().numpy(), d_ref.cpu().numpy()) < 1e-5)
    # small maps: the fused forms step aside (too few tiles per persistent workgroup / per chip)
    net = make_net('resnet50', {}, sd, 'bf16')
    net.set_profiling(True)
    net(x[:1, :256, :256].contiguous())
    small = {r['name']: r['kernel'] for r in net.get_profile()}
    assert not any('c3c1' in k or 'dual' in k for k in small.values()), small
    monkeypatch.setenv('DIRTORCH_AMD_C3C1', 'force')
    net = make_net('resnet50', {}, sd, 'bf16')
    net.set_profiling(True)
    d_forced = net(x[:2, :256, :256].contiguous())
    forced = {r['name']: r['kernel'] for r in net.get_profile()}
    assert forced.get('layer1.0.ds+c3c1') == 'conv_c3c1<64,ds>' and forced.get('layer2.2.c3c1') == 'conv_c3c1<128>'
    monkeypatch.setenv('DIRTORCH_AMD_C3C1', '0')
    d_plain = make_net('resnet50', {}, sd, 'bf16')(x[:2, :256, :256].contiguous())
    assert np.all(1 - O.cosine(d_forced.cpu().numpy(), d_plain.cpu().numpy()) < 1e-5)



@pytest.mark.parametrize('arch,B,H,W,dtype', [('resnet101', 1, 500, 375, 'fp16p'), ('resnet50', 8, 512, 512, 'bf16'),
                                             ('resnet101', 4, 1024, 1024, 'fp16'), ('resnet50', 3, 224, 224, 'fp16p'),
                                             ('resnet101', 1, 1024, 768, 'bf16')],
                         ids=['r101_b1_odd_splitk', 'r50_b8_512', 'r101_b4_1024', 'r50_b3_224', 'r101_b1_1024x768'])
def test_in_place_identity_blocks_are_bit_identical(arch, B, H, W, dtype, monkeypatch):
    """Since round 5 the identity blocks of layers 3-4 write their output over their input (engine.hip: every conv3 kernel
    reads the residual element it adds before it stores that element).  The descriptors and the trunk map must equal the
    ping-pong form's (DIRTORCH_AMD_NO_INPLACE) bit for bit - at batch 1 (split-K finalize, small tiles), at odd sizes (ragged
    tiles) and at the batched sizes where the register-stationary / persistent kernels run - and the FPN head, which keeps
    layer3's output, is unaffected."""
    import dir_oracle as O
    sd = O.synth_state_dict(arch, seed=7)
    g = torch.Generator(device='cuda').manual_seed(9)
    x = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8, device='cuda')
    monkeypatch.delenv('DIRTORCH_AMD_NO_INPLACE', raising=False)
    net = make_net(arch, {}, sd, dtype)
    d_in, f_in = net(x).clone(), net.forward_features(x).clone()
    net.set_profiling(True)
    net(x)
    kernels = {r['kernel'] for r in net.get_profile()}
    monkeypatch.setenv('DIRTORCH_AMD_NO_INPLACE', '1')
    ref = make_net(arch, {}, sd, dtype)
    d_pp, f_pp = ref(x), ref.forward_features(x)
    assert torch.equal(d_in, d_pp) and torch.equal(f_in, f_pp), (arch, B, H, W, dtype, sorted(kernels))
    assert torch.isfinite(d_in).all()


@pytest.mark.parametrize('stages', [3, 4])
def test_in_place_identity_blocks_with_paired_weights_beyond_layer1(stages, monkeypatch):
    """DIRTORCH_AMD_PAIR_STAGES >= 3 moves the paired-weight boundary into layers 3-4, whose identity blocks write in place:
    there conv3 runs on conv_pair.hip (residual fetched before the K loop, per tile) instead of the kernels the default
    mode's bit-identity test covers (round-5 advice).  Same gate: in place == ping-pong, bit for bit."""
    import dir_oracle as O
    sd = O.synth_state_dict('resnet50', seed=7)
    g = torch.Generator(device='cuda').manual_seed(19)
    x = torch.randint(0, 256, (2, 320, 256, 3), generator=g, dtype=torch.uint8, device='cuda')
    monkeypatch.setenv('DIRTORCH_AMD_PAIR_STAGES', str(stages))
    monkeypatch.delenv('DIRTORCH_AMD_NO_INPLACE', raising=False)
    net = make_net('resnet50', {}, sd, 'fp16p')
    d_in, f_in = net(x).clone(), net.forward_features(x).clone()
    net.set_profiling(True)
    net(x)
    used = {r['name']: r['kernel'] for r in net.get_profile()}
    assert used.get('layer3.2.conv3', '').startswith('conv_pair<'), used
    monkeypatch.setenv('DIRTORCH_AMD_NO_INPLACE', '1')
    ref = make_net('resnet50', {}, sd, 'fp16p')
    d_pp, f_pp = ref(x), ref.forward_features(x)
    assert torch.equal(d_in, d_pp) and torch.equal(f_in, f_pp)
    assert torch.isfinite(d_in).all()
