"""The CPU oracle (oracle/dir_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py from /root/reference)."""
import pickle

import numpy as np
import pytest
import torch

import dir_oracle as O

# mirrors MODEL_CASES of tests/golden/make_golden.py
CASES = [
    ('r18_gem', 'resnet18', dict(), 2.7, 2, 96, 80),
    ('r50_gem', 'resnet50', dict(), 2.7, 2, 97, 75),
    ('r50_gem_b1', 'resnet50', dict(), 3.0, 1, 64, 64),
    ('r50_max_norm', 'resnet50', dict(pooling='max', norm_features=True), None, 2, 64, 96),
    ('r50_avg_cb', 'resnet50', dict(pooling='avg', center_bias=0.5, out_dim=512), None, 2, 64, 64),
    ('r50_nofc', 'resnet50', dict(without_fc=True), 2.2, 2, 64, 64),
    ('r101_gem', 'resnet101', dict(), 2.7, 1, 128, 96),
]


def case_inputs(tag, arch, opts, gemp, B, H, W):
    sd = O.synth_state_dict(arch, seed=7, out_dim=opts.get('out_dim', 2048),
                            gemp=gemp if gemp else 3.0, pooling=opts.get('pooling', 'gem'))
    return sd, O.synth_images(11, B, H, W)


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_descriptor_matches_reference(case, model_goldens):
    tag, arch, opts, gemp, B, H, W = case
    sd, x = case_inputs(*case)
    kw = {k: v for k, v in opts.items() if k != 'out_dim'}
    with torch.no_grad():
        feat = O.resnet_features(sd, arch, x)
    desc = O.rmac_forward(sd, arch, x, **kw).numpy()
    gold = model_goldens[tag + '.desc']
    assert desc.shape == gold.shape            # (D,) when B == 1, like the reference's squeeze_
    assert tuple(model_goldens[tag + '.feat_shape']) == tuple(feat.shape)
    np.testing.assert_allclose(feat[:, ::64].numpy(), model_goldens[tag + '.feat_slice'],
                               rtol=1e-4, atol=1e-4)
    # same ops in the same order on the same CPU kernels: tolerance is thread-order noise only
    assert np.abs(desc - gold).max() < 2e-6
    assert np.all(O.cosine(desc, gold) > 1 - 1e-9)
    np.testing.assert_allclose(np.linalg.norm(desc.reshape(-1, desc.shape[-1]), axis=1), 1.0, atol=1e-5)


# mirrors HEAD_CASES of tests/golden/make_golden.py
HEAD_CASES = [
    ('r50_fpn', 'fpn', 'resnet50', dict(), 2, 97, 75),
    ('r18_fpn_norm', 'fpn', 'resnet18', dict(norm_features=True), 2, 96, 80),
    ('r50_fpn_b1', 'fpn', 'resnet50', dict(out_dim=512), 1, 64, 64),
    ('r50_fpn_nofc', 'fpn', 'resnet50', dict(without_fc=True), 2, 64, 96),
    ('r101_fpn0', 'fpn0', 'resnet101', dict(), 1, 128, 96),
    ('r50_cls', 'cls', 'resnet50', dict(out_dim=1000), 2, 64, 80),
    ('r18_cls_b1', 'cls', 'resnet18', dict(out_dim=256), 1, 64, 64),
]


def head_case_inputs(tag, head, arch, opts, B, H, W):
    feat = 512 * (4 if O.ARCH[arch][0] else 1)
    default_out = feat + feat // 2 if head in ('fpn', 'fpn0') else 2048
    sd = O.synth_state_dict(arch, seed=9, out_dim=opts.get('out_dim', default_out), gemp=2.6,
                            pooling='gem', head=head)
    return sd, O.synth_images(13, B, H, W)


def head_oracle(sd, head, arch, opts, x, quant=None):
    if head == 'cls':
        return O.classifier_forward(sd, arch, x, quant=quant)
    kw = {k: v for k, v in opts.items() if k != 'out_dim'}
    return O.fpn_forward(sd, arch, x, mode=1 if head == 'fpn' else 0, quant=quant, **kw)


@pytest.mark.parametrize('case', HEAD_CASES, ids=[c[0] for c in HEAD_CASES])
def test_fpn_and_classifier_match_reference(case, head_goldens):
    tag, head, arch, opts, B, H, W = case
    sd, x = head_case_inputs(*case)
    got = head_oracle(sd, head, arch, opts, x).numpy()
    gold = head_goldens[tag + '.desc']
    assert got.shape == gold.shape      # FPN squeezes at B == 1, the classifier keeps [1, out]
    scale = max(1.0, np.abs(gold).max())
    assert np.abs(got - gold).max() < 2e-6 * scale
    assert np.all(O.cosine(got, gold) > 1 - 1e-9)


def test_folded_bn_equals_unfolded():
    # the engine folds BatchNorm into the conv; the oracle's quant path does too - both must agree
    sd = O.synth_state_dict('resnet18', seed=3)
    x = O.synth_images(5, 1, 64, 64)
    a = O.rmac_forward(sd, 'resnet18', x).numpy()
    with torch.no_grad():
        fa = O.resnet_features(sd, 'resnet18', x)
        w, b = O._fold(sd, 'conv1.weight', 'bn1', None)
        y1 = torch.nn.functional.conv2d(x, w, b, 2, 3)
        y2 = O._conv_bn(sd, x, 'conv1.weight', 'bn1', 2, 3, None)
    np.testing.assert_allclose(y1.numpy(), y2.numpy(), rtol=1e-4, atol=1e-4)
    assert fa.shape == (1, 512, 2, 2) and a.shape == (2048,)


def test_quant_emulation_is_close():
    sd = O.synth_state_dict('resnet50', seed=7)
    x = O.synth_images(11, 2, 64, 64)
    ref = O.rmac_forward(sd, 'resnet50', x).numpy()
    for q, tol in (('bf16', 1e-4), ('fp16', 2e-6)):
        got = O.rmac_forward(sd, 'resnet50', x, quant=q).numpy()
        assert np.all(1 - O.cosine(got, ref) < tol), (q, 1 - O.cosine(got, ref))


def test_paired_head_emulations_are_ordered_by_what_they_pair():
    """The storage points of DIR_FP16P's two forms (oracle quant='fp16p': image, stem and layer1's 1x1 weights as fp16
    pairs; 'fp16pa': layer1's 3x3 weights and inner tensors too), on a BatchNorm-calibrated ResNet-50: each is closer to
    fp32 than the one before it, a BasicBlock net has no 1x1 in layer1 so both names mean the same thing there, and the
    pair representation itself holds ~22 bits."""
    import torch
    sd = O.calibrated_state_dict('resnet50', O.synth_images(99, 8, 96, 96), seed=7)
    x = O.synth_images(5, 4, 96, 96)
    ref = O.rmac_forward(sd, 'resnet50', x).numpy()
    err = {q if isinstance(q, str) else '%s/%d' % q: float((1 - O.cosine(O.rmac_forward(sd, 'resnet50', x, quant=q).numpy(), ref)).max())
           for q in ('fp16', 'fp16p', 'fp16pa', ('fp16pa', 2))}
    assert err['fp16pa/2'] < err['fp16pa'] < err['fp16p'] < err['fp16'], err
    sd18 = O.calibrated_state_dict('resnet18', O.synth_images(99, 8, 96, 96), seed=7)
    a = O.rmac_forward(sd18, 'resnet18', x, quant='fp16p')
    b = O.rmac_forward(sd18, 'resnet18', x, quant='fp16pa')
    assert torch.equal(a, b)
    v = torch.randn(4096, generator=torch.Generator().manual_seed(1)) * 3
    v = torch.where(v.abs() < 0.25, torch.full_like(v, 0.7), v)       # (tiny values: the lo plane goes subnormal)
    assert float(((O._q(v, 'pair') - v).abs() / v.abs()).max()) < 2.0 ** -21
    assert float(((O._q(v, 'fp16') - v).abs() / v.abs()).max()) > 2.0 ** -13


RESIZE_CASES = [(37, 53, 52, 75), (64, 64, 45, 45), (120, 90, 170, 127), (100, 100, 100, 141),
                (50, 70, 50, 35), (33, 47, 11, 13), (200, 300, 71, 424), (17, 19, 68, 76), (10, 10, 1, 1),
                (5, 7, 500, 3), (96, 128, 96, 128), (240, 320, 339, 452), (240, 320, 170, 226)]


@pytest.mark.parametrize('shape', RESIZE_CASES, ids=['%dx%d_to_%dx%d' % c for c in RESIZE_CASES])
def test_resize_restatement_is_bit_identical_to_pillow(shape):
    """The Scale transform (transforms.py:133-185) delegates to Pillow, which is the pin here."""
    from PIL import Image
    h, w, oh, ow = shape
    img = np.random.RandomState(h * 1000 + ow).randint(0, 256, (h, w, 3)).astype(np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    got = O.resize_bilinear_u8(img, ow, oh)
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_pool(postproc_goldens):
    g = postproc_goldens
    xs = [torch.from_numpy(a) for a in g['pool.in']]
    assert O.pool(xs[:1]) is xs[0]
    np.testing.assert_allclose(O.pool(xs, 'mean').numpy(), g['pool.mean'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(O.pool(xs, 'gem', 3).numpy(), g['pool.gem3'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(O.pool(xs, 'gem', 2.5).numpy(), g['pool.gem2.5'], rtol=1e-6, atol=1e-7)
    with pytest.raises(ValueError):
        O.pool(xs, 'median')


def test_whiten(postproc_goldens):
    g = postproc_goldens
    pca = O.PCAParams(g['pca.mean'], g['pca.components'], g['pca.var'], True)
    X = g['whiten.in']
    np.testing.assert_allclose(O.whiten_features(X, pca, whitenp=0.5), g['whiten.p0.5'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.whiten_features(X, pca, whitenp=0.25, whitenv=32, whitenm=2.0),
                               g['whiten.p0.25_v32_m2'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.whiten_features(X, pca, l2norm=False, whitenp=0.5), g['whiten.nol2'],
                               rtol=1e-5, atol=1e-5)


def test_fit_pca_matches_sklearn_subspace(postproc_goldens):
    # our SVD PCA spans the same components (up to sign) as the sklearn object in the golden file
    r = np.random.RandomState(0)
    X = r.standard_normal((200, 16)) @ np.diag(np.linspace(3, 0.5, 16))
    p = O.fit_pca(X)
    from sklearn.decomposition import PCA
    s = PCA(whiten=True).fit(X)
    np.testing.assert_allclose(p.explained_variance_, s.explained_variance_, rtol=1e-5)
    np.testing.assert_allclose(np.abs(p.components_ @ s.components_.T), np.eye(16), atol=1e-4)


def test_matmul(postproc_goldens):
    g = postproc_goldens
    np.testing.assert_allclose(O.matmul(g['matmul.A'], g['matmul.B']), g['matmul.np'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(O.matmul(g['matmul.A'], g['matmul.B']), g['matmul.torch'], rtol=1e-5, atol=1e-5)


def test_average_precision(postproc_goldens):
    g = postproc_goldens
    for ranks, val in zip(g['ap.ranks'], g['ap.values']):
        k = [int(v) for v in str(ranks).split(',') if v != '']
        assert O.compute_average_precision(np.array(k)) == pytest.approx(val, abs=1e-12)
    # derived known answers
    assert O.compute_average_precision([0, 1, 2]) == 1.0
    assert O.compute_average_precision([1, 3]) == pytest.approx(1 / 3)
    assert O.compute_average_precision([]) == 0.0


def test_eval_query_ap(postproc_goldens):
    g = postproc_goldens
    gnd = pickle.loads(g['evalap.gnd'][0])
    scores = g['evalap.scores']
    for q in range(scores.shape[0]):
        d = O.eval_query_AP(scores[q], gnd[q]['easy'], gnd[q]['hard'], gnd[q]['junk'])
        for mode in ('easy', 'medium', 'hard'):
            assert d[mode] == pytest.approx(g['evalap.' + mode][q], abs=1e-12)
    assert g['evalap.easy'][4] == -1
    m = O.mean_ap(scores, gnd)
    assert m['mAP-easy'] == pytest.approx(np.mean([v for v in g['evalap.easy'] if v >= 0]))


# mirrors QE_CASES of tests/golden/make_golden.py
QE_CASES = [(1, 0), (1, 3), (5, 0), (5, 3), (3, 1)]


def qe_inputs():
    import synth
    return synth.synth_descriptors(31, 12, 64), synth.synth_descriptors(32, 40, 64)


@pytest.mark.parametrize('k,alpha', QE_CASES)
def test_expand_descriptors_matches_reference(k, alpha, qe_goldens):
    """alpha-QE (db=) and DBA (self-set) of the oracle against the reference's expand_descriptors."""
    q, db = qe_inputs()
    got_self = O.expand_descriptors(db.copy(), alpha=alpha, k=k)
    got_db = O.expand_descriptors(q.copy(), db=db.copy(), alpha=alpha, k=k)
    np.testing.assert_allclose(got_self, qe_goldens['qe.self.k%d.a%d' % (k, alpha)], rtol=0, atol=1e-6)
    np.testing.assert_allclose(got_db, qe_goldens['qe.db.k%d.a%d' % (k, alpha)], rtol=0, atol=1e-6)
    assert O.expand_descriptors(q, db=db, alpha=alpha, k=0) is q
