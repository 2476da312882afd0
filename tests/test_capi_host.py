"""CPU-side checks: the C-ABI library loads and exports every symbol of include/dir_engine.h,
and the host-side mirror of dirtorch.nets behaves like the reference's factory.  No compute calls."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'dir_engine.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dir_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from dirtorch_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), 'missing export ' + s
        assert s in _lib.SIGNATURES, 'no ctypes signature for ' + s
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.dir_version().decode().endswith('gfx950')
    assert lib.dir_conv_variant_count() >= 4


def test_gpu_parity_matrix_mirror_of_the_admissibility_rule():
    """tests/test_ops_gpu.py generates its (shape, variant) matrix from its own mirror of conv_variant_admissible; the
    library's predicate (dir_conv_variant_admissible, host-only) must agree on every pair, so that no admissible pair goes
    untested and no inadmissible one is launched.  The default library ships no experiments kernel."""
    import ctypes
    import importlib.util
    from dirtorch_amd import _lib, ops
    spec = importlib.util.spec_from_file_location('ops_gpu_matrix', os.path.join(ROOT, 'tests', 'test_ops_gpu.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    names = ops.conv_variant_names()
    if 'DIRTORCH_AMD_LIB' not in os.environ:
        assert '128x256_ring1x1' not in names
    ok = ctypes.c_int()
    n_adm = 0
    for s in m.CONV_SHAPES:
        _, B, H, W, Cin, Cout, k, stride, pad, use_res, _ = s
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        for v, name in enumerate(names):
            _lib.call('dir_conv_variant_admissible', v, B, H, W, Cin, Cout, k, k, stride, pad, OH, OW, int(use_res), ctypes.byref(ok))
            assert bool(ok.value) == m.variant_admissible(name, Cin, Cout, k, stride, pad, use_res), (s[0], name, ok.value)
            n_adm += ok.value
    assert n_adm == len(m.CONV_CASES) and n_adm > 100
    with pytest.raises(_lib.DirError):
        _lib.call('dir_conv_variant_admissible', len(names), 1, 8, 8, 64, 64, 1, 1, 1, 0, 8, 8, 0, ctypes.byref(ok))


def test_switches_are_read_once_and_reloaded_on_request():
    """The DIRTORCH_AMD_* A/B switches are read from the environment ONCE (no getenv on any launch path); dir_reload_env
    re-reads them.  Observed through the host-only heuristic query: layer3's 3x3 takes the 512 x 128 patch kernel unless
    DIRTORCH_AMD_NO_PATCHW is set - and setting it has no effect until the reload."""
    import ctypes
    from dirtorch_amd import _lib
    lib = _lib.load()

    def pick():
        buf, ks = ctypes.create_string_buffer(64), ctypes.c_int()
        _lib.call('dir_conv_heuristic', 32, 64, 64, 256, 256, 3, 3, 1, 1, 64, 64, 0, buf, 64, ctypes.byref(ks))
        return buf.value.decode()
    saved = os.environ.pop('DIRTORCH_AMD_NO_PATCHW', None)
    try:
        assert lib.dir_reload_env() == 0
        assert pick() == '512x128_patch3x3w'
        os.environ['DIRTORCH_AMD_NO_PATCHW'] = '1'
        assert pick() == '512x128_patch3x3w'          # not re-read behind the host's back
        _lib.reload_env()
        assert pick() != '512x128_patch3x3w'
    finally:
        os.environ.pop('DIRTORCH_AMD_NO_PATCHW', None)
        if saved is not None:
            os.environ['DIRTORCH_AMD_NO_PATCHW'] = saved
        _lib.reload_env()
    assert pick() == '512x128_patch3x3w'


def test_every_switch_the_library_reads_is_documented():
    """Every DIRTORCH_AMD_* variable read_env() looks at (csrc/engine.hip) appears in the header's switch list or in
    INTEGRATION.md - a kernel switch nobody can find is a kernel nobody can bisect."""
    import re
    src = open(os.path.join(ROOT, 'deep-image-retrieval_amd', 'csrc', 'engine.hip')).read()
    body = src[src.index('static Env read_env()'):src.index('static Env g_env[2]')]
    names = set(re.findall(r'"(DIRTORCH_AMD_[A-Z0-9_]+)"', body))
    assert len(names) >= 25
    docs = open(os.path.join(ROOT, 'include', 'dir_engine.h')).read() + open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    missing = sorted(n for n in names if n not in docs and n.replace('DIRTORCH_AMD', '') not in docs)
    assert not missing, missing


def test_argument_errors_do_not_need_a_gpu():
    from dirtorch_amd import _lib
    lib = _lib.load()
    import ctypes
    rc = lib.dir_engine_set_tensor(None, b'x', None, None, 0)
    assert rc == -1 and b'null' in lib.dir_last_error()
    n = ctypes.c_int()
    assert lib.dir_engine_out_dim(None, ctypes.byref(n)) == -1
    with pytest.raises(_lib.DirError):
        _lib.call('dir_conv_variant_name', 9999, ctypes.create_string_buffer(8), 8)


def test_model_names_match_reference():
    from dirtorch_amd import nets
    expected = {'resnet18', 'resnet50', 'resnet101', 'resnet152',
                'resnet18_rmac', 'resnet50_rmac', 'resnet101_rmac', 'resnet152_rmac',
                'resnet18_fpn_rmac', 'resnet50_fpn_rmac', 'resnet101_fpn_rmac',
                'resnet101_fpn0_rmac', 'resnet152_fpn_rmac'}
    assert nets.model_names == expected        # dirtorch/nets/__init__.py:18-21 [probed: 13 names]
    with pytest.raises(NameError):
        nets.create_model('resnet34_rmac')
    with pytest.raises(ValueError):
        nets.create_model('resnet50_rmac', pooling='median')   # rmac_resnet.py:31
    # every name instantiates (host object only; no GPU needed until forward)
    fpn = nets.create_model('resnet50_fpn_rmac', scales=[1])
    assert fpn.out_dim == 3072 and fpn.state_dict()['fc.weight'].shape == (3072, 3072)
    assert fpn.state_dict()['conv1x5.weight'].shape == (1024, 2048, 1, 1)
    assert fpn.state_dict()['conv3c4.weight'].shape == (1024, 1024, 3, 3)
    fpn0 = nets.create_model('resnet101_fpn0_rmac')
    assert 'conv1x5.weight' not in fpn0.state_dict() and 'adpoolc4.p' in fpn0.state_dict()
    cls = nets.create_model('resnet18', out_dim=1000)
    assert cls.state_dict()['fc.weight'].shape == (1000, 512) and 'adpool.p' not in cls.state_dict()


def test_state_dict_keys_and_shapes_match_reference_layout():
    import dir_oracle as O
    from dirtorch_amd import nets
    for arch, nkeys in (('resnet50', 321), ('resnet101', 627)):   # SURVEY.md §5 [probed]
        net = nets.create_model(arch + '_rmac', scales=[1])       # 'scales' silently dropped
        sd = net.state_dict()
        assert len(sd) == nkeys
        ref = O.synth_state_dict(arch, seed=0)
        assert list(sd.keys()) == list(ref.keys())
        for k in sd:
            assert tuple(sd[k].shape) == tuple(ref[k].shape), k
        assert net.preprocess == dict(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225], input_size=224)
        net.load_state_dict({'module.' + k: v for k, v in ref.items()})   # DataParallel prefix
        assert torch.equal(net.state_dict()['adpool.p'], ref['adpool.p'])
        with pytest.raises(RuntimeError):
            net.load_state_dict({k: v for k, v in list(ref.items())[:10]})
        with pytest.raises(RuntimeError):
            bad = dict(ref)
            bad['fc.bias'] = torch.zeros(7)
            net.load_state_dict(bad)


def test_no_cpu_execution_path():
    from dirtorch_amd import nets
    net = nets.create_model('resnet18_rmac')
    with pytest.raises(RuntimeError):
        net.cpu()
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            net.cuda()
        with pytest.raises((RuntimeError, Exception)):
            net(torch.zeros(1, 3, 64, 64))


def test_tile_heuristic_choices_for_resnet101_at_1024():
    """The tile variant the engine picks for un-tuned shapes is what every default run uses (bench.py
    included).  Pin the choices for the ResNet-101 @ 1024^2 layer shapes at batch 32 and batch 1 - they
    were distilled from the autotuner and A/B runs on the GPU; a silent change here
    is a performance regression no numerics test would catch.  Pure host logic: runs without a GPU."""
    import ctypes
    from dirtorch_amd import _lib

    def pick(B, H, Cin, Cout, k, stride, res):
        pad = 1 if k == 3 else 0
        OH = (H + 2 * pad - k) // stride + 1
        buf, ks = ctypes.create_string_buffer(64), ctypes.c_int()
        _lib.call('dir_conv_heuristic', B, H, H, Cin, Cout, k, k, stride, pad, OH, OH, int(res), buf, 64,
                  ctypes.byref(ks))
        return buf.value.decode(), ks.value

    shapes = {   # name: (H of the input map, Cin, Cout, k, stride, residual)
        'l1.conv1': (256, 256, 64, 1, 1, 0), 'l1.conv2': (256, 64, 64, 3, 1, 0), 'l1.conv3': (256, 64, 256, 1, 1, 1),
        'l2.conv1': (128, 512, 128, 1, 1, 0), 'l2.conv2': (128, 128, 128, 3, 1, 0), 'l2.conv3': (128, 128, 512, 1, 1, 1),
        'l3.conv1': (64, 1024, 256, 1, 1, 0), 'l3.conv2': (64, 256, 256, 3, 1, 0), 'l3.conv3': (64, 256, 1024, 1, 1, 1),
        'l4.conv1': (32, 2048, 512, 1, 1, 0), 'l4.conv2': (32, 512, 512, 3, 1, 0), 'l4.conv3': (32, 512, 2048, 1, 1, 1),
    }
    batch32 = {
        'l1.conv1': '256x64_w4x1', 'l1.conv2': '256x64_patchlc3x3', 'l1.conv3': '256x64_w4x1',
        'l2.conv1': '256x128_w4x2_s3_k32', 'l2.conv2': '512x128_patch3x3w', 'l2.conv3': '64x512_wreg1x1',
        'l3.conv1': '256x256_persist1x1_x3', 'l3.conv2': '512x128_patch3x3w', 'l3.conv3': '64x512_wreg1x1',
        'l4.conv1': '256x256_persist1x1_x3', 'l4.conv2': '512x128_patch3x3w', 'l4.conv3': '256x256_persist1x1',
    }
    for name, s in shapes.items():
        assert pick(32, *s) == (batch32[name], 1), name
    batch1 = {    # small M: deep-ring small tiles, split-K where even those leave CUs idle
        'l2.conv2': ('64x128_w2x2_s4', 1), 'l3.conv1': ('64x64_w2x2_s4', 1), 'l3.conv2': ('64x64_w2x2_s4', 1),   # (round 6: one 64 x 64 tile per CU, 64 KB of LDS)
        'l3.conv3': ('128x128_w2x2', 1), 'l4.conv1': ('64x64_small_s4k2', 1), 'l4.conv2': ('64x128_w2x2', 8),   # (round 6, late: 128 tiles of 64 x 64 - conv_small.hip's two-K-steps-per-stage tile instead of split-K; K = 4608 keeps split-K)
    }
    for name, want in batch1.items():
        assert pick(1, *shapes[name]) == want, name
    # native-size images sit right under the 192-tile line (683 x 1024 -> 43 x 64 pixels in layer3; here 52^2 = 2 704 pixels, 172 tiles):
    # no split-K cliff there (round 6, from the tuner at batch 1: scripts/exp_batch1_tune.py, A/B on 1 / 2 / 4 streams gpurun_out/r6b1rules)
    assert pick(1, 52, 1024, 256, 1, 1, 0) == ('64x64_small_s4k2', 1) and pick(1, 52, 256, 256, 3, 1, 0) == ('64x64_small_s4k2', 1)
    assert pick(1, 28, 1024, 256, 1, 1, 0) == ('64x64_small_s4k2', 1)        # 52 tiles
    assert pick(1, 16, 1024, 256, 1, 1, 0)[0] != '64x64_small_s4k2'          # 16 tiles: split-K stays
    # between the two regimes (batch 4 here; ResNet-50 at 64 x 224^2 has the same pixel counts): 64 x 128 tiles, two workgroups per CU
    assert pick(4, *shapes['l3.conv1']) == ('64x128_w2x2', 1) and pick(4, *shapes['l3.conv2']) == ('64x128_w2x2', 1)
    assert pick(8, *shapes['l3.conv2']) == ('128x128_w2x2', 1)
    # long K loops in the small-map regime (round 6, from the tuner at batch 4 and ResNet-50 at 64 x 224^2: scripts/exp_tune_any.py)
    assert pick(4, *shapes['l4.conv2']) == ('128x128_w2x2', 4) and pick(64, 7, 512, 512, 3, 1, 0) == ('128x128_w2x2', 5)
    assert pick(4, *shapes['l4.conv1']) == ('64x128_w2x2_s4', 1) and pick(8, *shapes['l4.conv1']) == ('64x128_w2x2', 1)
    assert pick(8, *shapes['l4.conv2']) == ('64x128_w2x2', 1) and pick(2, *shapes['l4.conv2']) == ('64x64_w2x2_s4', 1)
    assert pick(64, 14, 1024, 512, 1, 1, 0) == ('128x128_w2x2', 1) and pick(64, 7, 512, 2048, 1, 1, 1) == ('128x128_w2x2', 1)   # config A's layer4: 392 / 400 tiles
    assert pick(2, 32, 512, 2048, 1, 1, 1) == ('64x128_w2x2', 1)                                                               # 256 tiles: the 64-pixel tile stays
    # ragged maps (configs[4]'s scales at batch 16): the patch tile wastes the rest of a map's last tiles, the flattened 16-wave tile
    # does not - the picker weighs tile fill x round fill of both (round 6, from the tuner: scripts/exp_multiscale_tune.py)
    assert pick(16, 107, 256, 256, 3, 1, 0) == ('256x256_w4x4', 1) and pick(16, 38, 512, 512, 3, 1, 0) == ('256x256_w4x4', 1)
    assert pick(16, 75, 256, 256, 3, 1, 0) == ('512x128_patch3x3w', 1) and pick(16, 54, 256, 256, 3, 1, 0) == ('512x128_patch3x3w', 1)
    assert pick(16, 54, 512, 512, 3, 1, 0) == ('512x128_patch3x3w', 1) and pick(16, 150, 128, 128, 3, 1, 0) == ('512x128_patch3x3w', 1)
    # ... and a map that fills a fraction of ONE tile per image is no reason to count that tile as a workgroup (config A's layer4: 64 x 7^2)
    assert pick(64, 7, 512, 512, 3, 1, 0)[0] != '512x128_patch3x3w' and pick(128, 7, 512, 512, 3, 1, 0)[0] != '512x128_patch3x3w'
    for B_ in (1, 4, 16, 64, 256):
        for H_ in (5, 7, 9, 14, 19, 27, 38):
            for C_ in (128, 256, 512):
                name_, _ = pick(B_, H_, C_, C_, 3, 1, 0)
                fill = H_ * H_ / float(-(-H_ // 16) * 16 * -(-H_ // 32) * 32)
                assert name_ != '512x128_patch3x3w' or fill >= 0.6 or B_ * H_ * H_ >= 192 * 512 // (C_ // 128), (B_, H_, C_, name_)
    # the register-stationary kernel needs a residual and enough pixel tiles per persistent workgroup
    # the strided 3x3 of layer2.0 (256^2 -> 128^2): the BK = 64 tile
    assert pick(32, 256, 128, 128, 3, 2, 0) == ('256x128_patchs2', 1)      # (round 6: the strided patch kernel)
    # ... only there: layer3.0 / 4.0's (256 / 512 channels) are matrix-bound and 10-15 % faster on the 16-wave tile (gpurun_out/r6s2b)
    assert pick(32, 128, 256, 256, 3, 2, 0) == ('256x256_w4x4', 1) and pick(32, 64, 512, 512, 3, 2, 0) == ('256x256_w4x4', 1)
    assert pick(1, 256, 128, 128, 3, 2, 0)[0] != '256x128_patchs2'           # 64 tiles: too few for the persistent kernel
    assert pick(32, 64, 256, 1024, 1, 1, 0)[0] != '64x512_wreg1x1'
    assert pick(2, 64, 256, 1024, 1, 1, 1)[0] != '64x512_wreg1x1'


def test_gemm_splitk_choices():
    """How many K slices the fp32 GEMM runs a shape with (host-only logic, dir_gemm_splitk_factor): the FC of a batch and
    PCA whitening / similarity of small sets split, anything with >= 128 output tiles or a short K does not, and a
    handful of Q rows stays on the one-wave-per-two-rows kernel."""
    from dirtorch_amd import _lib
    f = _lib.load().dir_gemm_splitk_factor
    assert f(2048, 32, 2048) == 16 and f(2048, 64, 2048) == 16 and f(2048, 256, 2048) == 8    # FC, batch 32 / 64 / 256
    assert f(4993, 70, 2048) == 6                                  # ROxford-size similarity: 40 tiles
    assert f(2048, 1, 2048) == 1 and f(2048, 4, 2048) == 1         # batch 1-4: the row-per-wave kernel
    assert f(1006322, 70, 2048) == 1 and f(2048, 100000, 2048) == 1   # plenty of tiles
    assert f(2048, 32, 256) == 1                                   # 8 slabs of K: not worth a second kernel
    for NP, NQ, K in ((2048, 512, 4096), (300, 70, 1031), (515, 33, 2080)):
        s = f(NP, NQ, K)
        assert 1 <= s <= 16 and s * NQ * NP * 4 <= 64 << 20 and (s == 1 or (K + 31) // 32 // s >= 4)


def test_profile_tooling_knows_every_engine_kernel():
    """scripts/summarize_prof.py aligns rocprofv3 dispatches with bench.py's launch records BY NAME; a kernel it cannot
    name silently drops out of the per-kernel roofline table (this happened to two new kernels in round 2).  Take the
    kernel names from the library itself (host-side launch stubs) and require: every convolution kernel maps onto a
    registered tile-variant name or a fused-seam name, every other kernel of the forward path onto the name bench.py's
    profile records carry."""
    import re
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import summarize_prof as S
    from dirtorch_amd import _lib, ops
    so = os.path.join(ROOT, 'deep-image-retrieval_amd', 'dirtorch_amd', 'libdir_engine.so')
    _lib.load()
    raw = subprocess.run('strings -n 8 %s | grep -E "^_Z.*__device_stub__" | sort -u | c++filt' % so, shell=True,
                         capture_output=True, text=True, check=True).stdout.splitlines()
    names = sorted({re.sub(r'\(.*$', '', l).replace('__device_stub__', '').replace('void ', '') for l in raw})
    assert len(names) > 100, names[:5]
    variants = set(ops.conv_variant_names())
    forward = {'stem_pool', 'prep_input', 'global_pool', 'gemm_nt_f32', 'maxpool_3x3s2', 'upsample_add',
               'prep_input_f32', 'maxpool_f32', 'global_pool_f32', 'upsample_add_f32',      # + the strict path's
               'stem_pool_pair', 'prep_input_pair',                                        # + the paired head's
               'stem_pool_u8', 'prep_input_u8'}                                            # + its uint8-feed form (stem_u8.hip)
    other = {'l2norm_rows_kernel', 'multiscale_pool_kernel', 'rank_sort_kernel', 'rank_hist_kernel', 'rank_finalize_kernel', 'revisitop_ap_kernel', 'expand_rows_kernel',
             'resample_coeffs_kernel', 'resample_pass_kernel', 'sim_split_kernel', 'sim_split_lc_kernel', 'split_queries_kernel', 'whiten_split_kernel', 'fill_noise_kernel',
             'gemm_splitk_finalize_kernel', 'conv_splitk_finalize_kernel', 'conv_naive_kernel',
             'pack_patchs2_kernel', 'pack_patchw_kernel'}   # (filter re-ordering at finalize / in the per-op entry points)
    for n in names:
        k = S.bench_kernel_name(S.short(n))
        if 'conv' in n and 'finalize' not in n and 'naive' not in n:
            m = re.match(r'^conv_igemm<([^>/]+)(/splitk|/dual)?>$', k)
            assert (m and m.group(1) in variants) or re.match(r'^conv_c3c1<(64|128)(,ds)?(,wp)?>$', k) or k == 'conv_seam3<256>' or \
                re.match(r'^conv_f32<128x(64|128)>$', k) or re.match(r'^conv_pair<128x(64|128)_(patch3x3_)?x?w(/dual)?>$', k), (n, k)
        else:
            assert k in forward or k in other, (n, k)
