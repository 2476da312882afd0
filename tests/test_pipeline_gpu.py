"""Pipeline- and CLI-level parity on the MI355X (SURVEY.md §4 levels 4-5):
the drop-in entry points of dirtorch_amd (test_dir / extract_features) against the CPU oracle run
over the same files, checkpoint (incl. a pickled sklearn PCA) and revisitop-format ground truth."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def save_images(root, names, sizes, seed):
    from PIL import Image
    r = np.random.RandomState(seed)
    os.makedirs(root, exist_ok=True)
    for name, (h, w) in zip(names, sizes):
        yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing='ij')
        img = np.stack([127 + 100 * np.sin(r.uniform(2, 9) * yy * 6.28 + r.uniform(0, 6)) *
                        np.cos(r.uniform(2, 9) * xx * 6.28) + 20 * r.standard_normal((h, w)) for _ in range(3)], -1)
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(root, name))


def oracle_descriptors(sd, arch, files, rois=None, quant=None):
    """What the reference computes per image: PIL RGB (-> crop) -> ToTensor -> Normalize -> net.
    quant='bf16'|'fp16': the same with the engine's 16-bit storage points emulated (an IDEAL 16-bit
    implementation; used to derive what a dtype can lose, never as the thing compared against)."""
    import dir_oracle as O
    from PIL import Image
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    out = []
    for i, f in enumerate(files):
        img = Image.open(f).convert('RGB')
        if rois is not None:
            img = img.crop(rois[i])
        x = (torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float() / 255 - mean) / std
        out.append(O.rmac_forward(sd, arch, x[None], quant=quant).reshape(1, -1))
    return torch.cat(out, 0)


def make_checkpoint(path, arch, sd, pca):
    torch.save({'model_options': dict(arch=arch + '_rmac', out_dim=2048, pooling='gem', gemp=3),
                'state_dict': {'module.' + k: v for k, v in sd.items()},     # DataParallel-style keys
                'pca': {'Landmarks_clean': pca}, 'epoch': 3}, path)


def fitted_pca(seed=0, n=96, d=2048):
    from sklearn.decomposition import PCA
    r = np.random.RandomState(seed)
    base = r.standard_normal((n, 24)).astype(np.float32) @ r.standard_normal((24, d)).astype(np.float32)
    base += 0.05 * r.standard_normal((n, d)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    return PCA(n_components=32, whiten=True).fit(base)


def test_extract_features_cli(tmp_path):
    """python -m dirtorch_amd.extract_features on an ImageList with whitening -> .npy"""
    import dir_oracle as O
    from dirtorch_amd import extract_features as ef
    names = ['a.png', 'b.png', 'c.png', 'd.png', 'e.png']
    sizes = [(96, 128), (130, 90), (64, 64), (101, 77), (80, 144)]        # variable H x W, odd sizes
    save_images(str(tmp_path / 'imgs'), names, sizes, 1)
    (tmp_path / 'list.txt').write_text('\n'.join(names) + '\n')
    sd = O.synth_state_dict('resnet18', seed=7, gemp=3.0)
    pca = fitted_pca()
    ck = str(tmp_path / 'synth.pt')
    make_checkpoint(ck, 'resnet18', sd, pca)
    out = str(tmp_path / 'out' / 'feats.npy')
    ef.main(['--dataset', 'ImageList("%s", root="%s")' % (tmp_path / 'list.txt', tmp_path / 'imgs'),
             '--checkpoint', ck, '--output', out, '--gpu', '0', '--threads', '2',
             '--whiten', 'Landmarks_clean', '--whitenp', '0.5'])
    got = np.load(out)
    ref = oracle_descriptors(sd, 'resnet18', [str(tmp_path / 'imgs' / n) for n in names]).numpy()
    ref_w = O.whiten_features(ref, O.PCAParams(pca.mean_, pca.components_, pca.explained_variance_, True), whitenp=0.5)
    assert got.shape == ref_w.shape == (5, 32)
    # whitening subtracts the mean and rescales: it amplifies upstream error, so this is the strict gate
    assert np.all(1 - O.cosine(got, ref_w) < 1e-4), 1 - O.cosine(got, ref_w)
    # un-whitened descriptors too
    out2 = str(tmp_path / 'raw.npy')
    ef.main(['--dataset', 'ImageList("%s", root="%s")' % (tmp_path / 'list.txt', tmp_path / 'imgs'),
             '--checkpoint', ck, '--output', out2, '--gpu', '0', '--threads', '0'])
    raw = np.load(out2)
    assert raw.shape == (5, 2048) and np.all(1 - O.cosine(raw, ref) < 1e-4)


@pytest.mark.parametrize('dtype', ['fp16', 'bf16'])
def test_test_dir_cli_roxford_protocol(tmp_path, monkeypatch, dtype):
    """python -m dirtorch_amd.test_dir --dataset ROxford5K on a synthetic revisitop-format dataset
    (files, ROI crops, pickled PCA, JSON output, --save-feats).  Three checks:
      1. the saved descriptors against the fp32 oracle: 1e-4 cosine in fp16 (north-star); in bf16 three
         times what an ideal bf16-storage implementation loses on the same files (quant= emulation);
      2. everything AFTER the descriptors is exact: the CLI's mAPs equal the oracle's whitening +
         similarity + AP protocol run on the CLI's own descriptors;
      3. fp16: mAP within 0.1 point (1e-3) of the fp32 oracle end to end.  (bf16 gets no end-to-end mAP
         number here: with 14 images one rank flip moves a mode's mAP by 1.5e-3, the smallest score gap
         is 8e-4 and even an ideal bf16 implementation perturbs scores by 1.7e-2; its statistically
         meaningful mAP gate is test_extract_whiten_rank_map_parity.)"""
    import dir_oracle as O
    from dirtorch_amd import test_dir as td
    monkeypatch.setenv('DIRTORCH_AMD_DTYPE', dtype)
    root = tmp_path / 'oxford5k'
    N, Q = 14, 3
    r = np.random.RandomState(5)
    names = ['im%02d' % i for i in range(N)]
    sizes = [(int(r.randint(70, 130)), int(r.randint(70, 130))) for _ in range(N)]
    save_images(str(root / 'jpg'), [n + '.jpg' for n in names], sizes, 2)
    # planted structure: positives are noisy copies of the query image
    from PIL import Image
    gnd = []
    for q in range(Q):
        base = np.asarray(Image.open(str(root / 'jpg' / (names[q] + '.jpg'))).convert('RGB')).astype(np.float32)
        h, w = sizes[q]
        easy = [3 + 3 * q, 4 + 3 * q]
        hard = [5 + 3 * q]
        for j, sigma in zip(easy + hard, (4.0, 8.0, 25.0)):
            noisy = np.clip(base + sigma * r.standard_normal(base.shape), 0, 255).astype(np.uint8)
            Image.fromarray(noisy).save(str(root / 'jpg' / (names[j] + '.jpg')), quality=95)
            sizes[j] = (h, w)
        gnd.append({'bbx': [4, 6, w - 5, h - 3], 'easy': easy, 'hard': hard, 'junk': [q]})
    with open(str(root / 'gnd_roxford5k.pkl'), 'wb') as f:
        pickle.dump({'imlist': names, 'qimlist': names[:Q], 'gnd': gnd}, f)
    monkeypatch.setenv('DB_ROOT', str(tmp_path))
    sd = O.synth_state_dict('resnet18', seed=7, gemp=3.0)
    files = [str(root / 'jpg' / (n + '.jpg')) for n in names]
    rois = [tuple(g['bbx']) for g in gnd]
    bd = oracle_descriptors(sd, 'resnet18', files).numpy()
    qd = oracle_descriptors(sd, 'resnet18', files[:Q], rois).numpy()
    # whitening learned on the descriptor distribution itself (as Landmarks_clean is for the real
    # models): a PCA unrelated to the data would turn the ranking into noise amplification
    from sklearn.decomposition import PCA
    pca = PCA(n_components=8, whiten=True).fit(bd)
    ck = str(tmp_path / 'synth.pt')
    make_checkpoint(ck, 'resnet18', sd, pca)
    js = str(tmp_path / 'res' / 'out.json')
    feats = str(tmp_path / 'feats')
    res = td.main(['--dataset', 'ROxford5K', '--checkpoint', ck, '--gpu', '0', '--threads', '2',
                   '--whiten', 'Landmarks_clean', '--whitenp', '0.25', '--out-json', js, '--detailed',
                   '--save-feats', feats])
    P = O.PCAParams(pca.mean_, pca.components_, pca.explained_variance_, True)

    def protocol_map(b, q):
        return O.mean_ap(O.matmul(O.whiten_features(q, P, whitenp=0.25), O.whiten_features(b, P, whitenp=0.25)), gnd)

    got_b = np.load(os.path.join(feats, 'feats.bdescs.npy'))
    got_q = np.load(os.path.join(feats, 'feats.qdescs.npy'))
    e_raw = max((1 - O.cosine(got_b, bd)).max(), (1 - O.cosine(got_q, qd)).max())
    allow = 1e-4
    if dtype == 'bf16':
        eb = oracle_descriptors(sd, 'resnet18', files, quant='bf16').numpy()
        eq = oracle_descriptors(sd, 'resnet18', files[:Q], rois, quant='bf16').numpy()
        allow = max(1e-4, 3 * max((1 - O.cosine(eb, bd)).max(), (1 - O.cosine(eq, qd)).max()))
    ref, own = protocol_map(bd, qd), protocol_map(got_b, got_q)
    print('\n[pipeline-cli] %s: descriptors 1-cos %.2e (allowance %.2e); mAP easy/medium/hard engine %s, oracle on the '
          "engine's descriptors %s, fp32 oracle %s" % (dtype, e_raw, allow, [round(res[k], 5) for k in sorted(ref)],
                                                       [round(own[k], 5) for k in sorted(ref)], [round(ref[k], 5) for k in sorted(ref)]))
    assert e_raw < allow, (e_raw, allow)
    for k in ('mAP-easy', 'mAP-medium', 'mAP-hard'):
        assert abs(res[k] - own[k]) < 1e-9, (k, res[k], own[k])
        if dtype == 'fp16':
            assert abs(res[k] - ref[k]) < 1e-3, (k, res[k], ref[k])
    assert len(res['APs-medium']) == Q and os.path.isfile(js)


@pytest.mark.parametrize('ckpt', ['synthetic', 'calibrated'])
@pytest.mark.parametrize('dtype', ['fp16', 'fp16p', 'bf16', 'f32'])
def test_extract_whiten_rank_map_parity(dtype, ckpt):
    """extraction -> PCA whitening -> similarity -> revisitop mAP on 400 images / 25 queries with
    planted near-duplicates, on two checkpoints:

      synthetic   random He-init weights: descriptors of unrelated images are collinear (cosine 0.9998),
                  so whitening (mean subtraction + 1/sigma^0.25) amplifies trunk rounding noise ~70x;
      calibrated  the same weights with BatchNorm statistics calibrated on synthetic images
                  (tests/synth.py): unrelated descriptors have cosine ~0.8, the regime of a trained net.

    Gates.  fp16: the north-star tolerances as stated - descriptors and whitened descriptors within
    1e-4 cosine, mAP within 0.1 point (1e-3) - on the calibrated checkpoint; on the collinear one the
    whitened gate is what an ideal fp16 implementation achieves there (emulation-derived, as below).
    bf16: no fixed number is honest - an IDEAL bf16-storage implementation (the oracle's quant=
    emulation of the engine's storage points) is itself 2e-4 / 1e-3 away on the calibrated checkpoint -
    so every bf16 allowance is 3 x the emulation's own distance from fp32, computed here on the same
    data, and the engine must also sit within that distance OF the emulation.
    fp16p (fp16 with the paired head, conv_pair.hip): gated like fp16 - the stated numbers on the calibrated checkpoint,
    which it meets with a ~10x margin where fp16 has none (tests/test_pair_gpu.py holds it to them at the BASELINE sizes).
    f32 (the strict path, conv_f32.hip): the stated numbers on BOTH checkpoints, no allowance of any kind."""
    import dir_oracle as O
    from dirtorch_amd import nets
    from dirtorch_amd.utils import common
    r = np.random.RandomState(11)
    N, Q, S = 400, 25, 64
    imgs = O.synth_images(21, N, S, S).numpy()
    gnd = []
    for q in range(Q):
        idx = r.choice(np.arange(Q, N), 12, replace=False)
        for j, sigma in zip(idx[:8], (0.05, 0.08, 0.1, 0.15, 0.3, 0.4, 0.5, 0.6)):
            imgs[j] = imgs[q] + sigma * r.standard_normal(imgs[q].shape).astype(np.float32)
        gnd.append({'easy': sorted(idx[:4].tolist()), 'hard': sorted(idx[4:8].tolist()),
                    'junk': sorted([q] + idx[8:].tolist())})
    x = torch.from_numpy(imgs)
    if ckpt == 'synthetic':
        sd = O.synth_state_dict('resnet18', seed=7)
    else:
        sd = O.calibrated_state_dict('resnet18', O.synth_images(99, 32, S, S), seed=7)

    def oracle(quant=None):
        return torch.cat([O.rmac_forward(sd, 'resnet18', x[i:i + 50], quant=quant) for i in range(0, N, 50)]).numpy()

    ref = oracle()
    emu = ref if dtype == 'f32' else oracle(dtype)
    net = nets.create_model('resnet18_rmac', pretrained='')
    net.load_state_dict(sd)
    net.compute_dtype = dtype
    net.cuda()
    got = torch.cat([net(x[i:i + 50].cuda()) for i in range(0, N, 50)]).cpu().numpy()
    pair = ref[Q:Q + 100] @ ref[Q + 100:Q + 200].T
    if ckpt == 'calibrated':
        assert pair.mean() < 0.9, pair.mean()               # the checkpoint is not rank-collapsed
    P = O.fit_pca(ref[Q:])
    kw = dict(whitenp=0.25, whitenv=16)
    ref_w, emu_w = O.whiten_features(ref, P, **kw), O.whiten_features(emu, P, **kw)
    got_w = common.whiten_features(got, P, **kw)
    # the whitening kernel itself adds nothing: same input, device vs oracle
    assert np.all(1 - O.cosine(common.whiten_features(ref, P, **kw), ref_w) < 1e-6)
    e_raw, e_w = (1 - O.cosine(got, ref)).max(), (1 - O.cosine(got_w, ref_w)).max()
    i_raw, i_w = (1 - O.cosine(emu, ref)).max(), (1 - O.cosine(emu_w, ref_w)).max()
    m_ref = O.mean_ap(O.matmul(ref_w[:Q], ref_w), gnd)
    m_emu = O.mean_ap(O.matmul(emu_w[:Q], emu_w), gnd)
    m_got = O.mean_ap(common.matmul(got_w[:Q], got_w), gnd)
    d_map = max(abs(m_ref[k] - m_got[k]) for k in m_ref)
    i_map = max(abs(m_ref[k] - m_emu[k]) for k in m_ref)
    print('\n[pipeline] %s %s: mean cosine of unrelated images %.4f | 1-cos raw: engine %.2e ideal-16bit %.2e | '
          'whitened: engine %.2e ideal %.2e | max |dmAP|: engine %.2e ideal %.2e | mAP-medium %.3f'
          % (ckpt, dtype, pair.mean(), e_raw, i_raw, e_w, i_w, d_map, i_map, m_ref['mAP-medium']))
    if dtype == 'f32' or (dtype in ('fp16', 'fp16p') and ckpt == 'calibrated'):   # the north-star numbers, as stated
        assert e_raw < 1e-4 and e_w < 1e-4 and d_map < 1e-3, (e_raw, e_w, d_map)
    else:
        assert e_raw < max(1e-4, 3 * i_raw), (e_raw, i_raw)
        assert e_w < max(1e-4, 3 * i_w), (e_w, i_w)
        assert d_map < 1e-3 + 3 * i_map, (d_map, i_map)
    # and the engine is an implementation OF that 16-bit arithmetic: as close to the emulation as the
    # emulation is to fp32
    assert (1 - O.cosine(got, emu)).max() < max(1e-5, 4 * i_raw)   # two independent roundings add: ~2x
    assert m_ref['mAP-easy'] > 0.5        # the planted structure is actually retrievable


def test_multiscale_device_resize_equals_cpu_pil_path(tmp_path, monkeypatch):
    """--trfs with several Scale(..) chains: the fused path (decode + upload once, Pillow-identical
    resize on the GPU) must give the very same file as one PIL-resizing loader pass per scale, and
    both must match the oracle run over PIL-resized images + common.pool (test_dir.py:118-122)."""
    import dir_oracle as O
    from PIL import Image
    from dirtorch_amd import extract_features as ef
    names = ['a.png', 'b.png', 'c.png', 'd.png']
    sizes = [(96, 128), (130, 90), (75, 101), (64, 112)]
    save_images(str(tmp_path / 'imgs'), names, sizes, 3)
    (tmp_path / 'list.txt').write_text('\n'.join(names) + '\n')
    sd = O.synth_state_dict('resnet18', seed=7, gemp=3.0)
    ck = str(tmp_path / 'synth.pt')
    make_checkpoint(ck, 'resnet18', sd, fitted_pca())
    monkeypatch.setenv('DIRTORCH_AMD_DTYPE', 'fp16')
    chains = ['', 'Scale(1.414)', 'Scale(0.707)']
    outs = {}
    for pooling in ('mean', 'gem'):
        for flag in ('1', '0'):
            monkeypatch.setenv('DIRTORCH_AMD_DEVICE_SCALE', flag)
            out = str(tmp_path / ('ms_%s_%s.npy' % (pooling, flag)))
            argv = ['--dataset', 'ImageList("%s", root="%s")' % (tmp_path / 'list.txt', tmp_path / 'imgs'),
                    '--checkpoint', ck, '--output', out, '--gpu', '0', '--threads', '0', '--pooling', pooling,
                    '--gemp', '3', '--trfs'] + chains
            ef.main(argv)
            outs[pooling, flag] = np.load(out)
        # bit-identical resize -> bit-identical descriptors
        assert np.array_equal(outs[pooling, '1'], outs[pooling, '0'])

    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    per_scale = []
    for f in (0, 1.414, 0.707):
        rows = []
        for n in names:
            img = Image.open(str(tmp_path / 'imgs' / n)).convert('RGB')
            if f:
                img = img.resize((int(0.5 + f * img.size[0]), int(0.5 + f * img.size[1])), Image.BILINEAR)
            x = (torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float() / 255 - mean) / std
            rows.append(O.rmac_forward(sd, 'resnet18', x[None]).reshape(1, -1))
        per_scale.append(torch.cat(rows, 0))
    # the signed cube root of 'gem' is ill-conditioned where the scales cancel (d/dx x^(1/3) -> inf
    # at 0), so fp16-level input error shows up as ~1e-4; 'mean' is the tight gate
    for pooling, tol in (('mean', 1e-4), ('gem', 2e-3)):
        ref = torch.nn.functional.normalize(O.pool(per_scale, pooling, 3), dim=1).numpy()
        err = 1 - O.cosine(outs[pooling, '1'], ref)
        assert np.all(err < tol), (pooling, err)


def test_bucketed_batching_keeps_order_and_values(tmp_path, monkeypatch):
    """Variable-size images batched by size on the GPU (test_dir._extract_bucketed) against the
    reference's batch-1 loop: same rows in the same order; values equal up to the split-K
    re-association that batch 1 triggers in the deep layers."""
    import dir_oracle as O
    from dirtorch_amd import datasets, nets
    from dirtorch_amd import test_dir as td
    sizes = [(96, 128), (130, 90), (96, 128), (64, 64), (96, 128), (130, 90), (96, 128), (80, 144), (96, 128),
             (64, 64), (130, 90)]
    names = ['im%02d.png' % i for i in range(len(sizes))]
    save_images(str(tmp_path / 'imgs'), names, sizes, 5)
    (tmp_path / 'list.txt').write_text('\n'.join(names) + '\n')
    db = datasets.create('ImageList("%s", root="%s")' % (tmp_path / 'list.txt', tmp_path / 'imgs'))
    net = nets.create_model('resnet18_rmac', pretrained='')
    net.load_state_dict(O.synth_state_dict('resnet18', seed=7, gemp=3.0))
    net.compute_dtype = 'fp16'
    net.cuda().eval()
    monkeypatch.setenv('DIRTORCH_AMD_BUCKET_BATCH', '1')
    a = td.extract_image_features(db, '', net, threads=0, batch_size=3).cpu()
    monkeypatch.setenv('DIRTORCH_AMD_BUCKET_BATCH', '0')
    b = td.extract_image_features(db, '', net, threads=0, batch_size=3).cpu()
    assert a.shape == b.shape == (len(sizes), 2048)
    assert float((1 - (a * b).sum(1)).max()) < 1e-6
    # identical images (none here) aside, rows must not be permuted: every row is closest to itself
    assert torch.equal((a @ b.t()).argmax(1), torch.arange(len(sizes)))
    ref = oracle_descriptors(O.synth_state_dict('resnet18', seed=7, gemp=3.0), 'resnet18',
                             [str(tmp_path / 'imgs' / n) for n in names])
    assert float((1 - (a * ref).sum(1)).max()) < 1e-4


def test_eval_model_with_query_expansion_and_db_augmentation(tmp_path, monkeypatch):
    """test_dir.eval_model(..., aqe=, adba=) from saved features (--load-feats): whitening -> alpha-DBA of the
    database -> alpha-QE of the queries against the augmented database -> similarity -> AP, every step on the
    GPU, against the oracle's restatement of the same chain (expand_descriptors pinned to the reference's
    outputs in tests/test_oracle_golden.py).  The reference reads `args.adba / args.aqe` from a module global
    inside eval_model (test_dir.py:141,143); the function parameters are what is meant."""
    import dir_oracle as O
    import synth
    from dirtorch_amd import datasets
    from dirtorch_amd import test_dir as td
    N, Q, D = 300, 12, 256
    r = np.random.RandomState(3)
    gnd = []
    for q in range(Q):
        idx = r.choice(np.arange(Q, N), 14, replace=False)
        gnd.append({'bbx': [0, 0, 1, 1], 'easy': sorted(idx[:5].tolist()), 'hard': sorted(idx[5:10].tolist()),
                    'junk': sorted(idx[10:].tolist())})
    f = str(tmp_path / 'gnd.pkl')
    with open(f, 'wb') as fh:
        pickle.dump({'imlist': ['i%d' % i for i in range(N)], 'qimlist': ['q%d' % i for i in range(Q)], 'gnd': gnd}, fh)
    db = datasets.ImageListRelevants(f, root=str(tmp_path))
    bdescs = synth.synth_descriptors(51, N, D, clusters=20)
    qdescs = synth.synth_descriptors(52, Q, D, clusters=20)
    for q in range(Q):                      # plant the positives near their query
        for j in gnd[q]['easy'] + gnd[q]['hard']:
            v = bdescs[j] * 0.6 + qdescs[q]
            bdescs[j] = v / np.linalg.norm(v)
    feats = tmp_path / 'feats'
    feats.mkdir()
    np.save(str(feats / 'feats.bdescs.npy'), bdescs)
    np.save(str(feats / 'feats.qdescs.npy'), qdescs)
    pca = O.fit_pca(bdescs)

    class Net(object):                      # eval_model only reads net.pca on the --load-feats path
        pass
    net = Net()
    net.pca = pca
    whiten = dict(whitenp=0.25, whitenv=64, whitenm=1.0)
    res = td.eval_model(db, net, '', whiten=whiten, aqe={'k': 3, 'alpha': 2}, adba={'k': 4, 'alpha': 1},
                        load_feats=str(feats), detailed=True)
    b = O.whiten_features(bdescs, pca, **whiten)
    q = O.whiten_features(qdescs, pca, **whiten)
    b = O.expand_descriptors(b, alpha=1, k=4)
    q = O.expand_descriptors(q, db=b, alpha=2, k=3)
    ref = O.mean_ap(O.matmul(q, b), gnd)
    for k in ref:
        assert abs(res[k] - ref[k]) < 1e-9, (k, res[k], ref[k])
    plain = td.eval_model(db, net, '', whiten=whiten, load_feats=str(feats))
    assert any(abs(plain[k] - res[k]) > 1e-6 for k in ref)      # the expansion actually changed the ranking


def test_fp16_overflow_is_reported_not_returned(tmp_path):
    """Activations that leave the fp16 range in the last stage turn into inf/NaN descriptors; the extraction
    loops must say so (and name the bf16 switch) instead of handing them on.  A checkpoint with a huge
    BatchNorm gain in the last block does it.  (Deep inside the trunk an overflow need not surface: the
    hardware max of the fused ReLU returns the non-NaN operand, so inf - inf = NaN becomes 0 one layer later;
    the guard is a tripwire for the common case, not a proof - DESIGN.md section 4.)"""
    import dir_oracle as O
    from dirtorch_amd import datasets, nets
    from dirtorch_amd import test_dir as td
    names = ['a.png', 'b.png']
    save_images(str(tmp_path / 'imgs'), names, [(64, 64), (64, 64)], 7)
    (tmp_path / 'list.txt').write_text('\n'.join(names) + '\n')
    db = datasets.create('ImageList("%s", root="%s")' % (tmp_path / 'list.txt', tmp_path / 'imgs'))
    sd = O.synth_state_dict('resnet18', seed=7, gemp=3.0)
    sd['layer4.1.bn2.weight'] = sd['layer4.1.bn2.weight'] * 1e6
    for dtype, ok in (('fp16', False), ('bf16', True)):
        net = nets.create_model('resnet18_rmac', pretrained='')
        net.load_state_dict(sd)
        net.compute_dtype = dtype
        net.cuda().eval()
        if ok:
            d = td.extract_image_features(db, '', net, threads=0)
            assert torch.isfinite(d).all()
        else:
            with pytest.raises(FloatingPointError) as ei:
                td.extract_image_features(db, '', net, threads=0)
            assert 'DIRTORCH_AMD_DTYPE=bf16' in str(ei.value)


def test_fp16_overflow_hidden_by_a_later_relu_is_still_reported(tmp_path):
    """An overflow planted in layer2: a huge bn1 shift makes conv1's post-ReLU output +inf in fp16, conv2 then
    sums inf - inf = NaN, and its fused ReLU - a hardware max, which returns the non-NaN operand - flushes
    the NaNs to 0: the descriptors come out FINITE and wrong.  The engine's overflow word (every kernel that
    stores an fp16 inf / NaN sets it, dir_engine_overflow) must still turn that into an error; bf16 (fp32's
    exponent range) runs clean.  The reference computes in fp32 and cannot overflow
    (dirtorch/nets/backbones/resnet.py:67-87)."""
    import dir_oracle as O
    from dirtorch_amd import datasets, nets
    from dirtorch_amd import test_dir as td
    names = ['a.png', 'b.png']
    save_images(str(tmp_path / 'imgs'), names, [(96, 96), (96, 96)], 7)
    (tmp_path / 'list.txt').write_text('\n'.join(names) + '\n')
    db = datasets.create('ImageList("%s", root="%s")' % (tmp_path / 'list.txt', tmp_path / 'imgs'))
    x = O.synth_images(5, 2, 96, 96).cuda()
    for arch, key in (('resnet50', 'layer2.0.bn1.bias'), ('resnet18', 'layer2.0.bn1.bias')):
        sd = O.synth_state_dict(arch, seed=7, gemp=3.0)
        sd[key] = sd[key] + 1e5          # (a shift, not a gain: the folded WEIGHTS stay in range, see the last check)
        for dtype in ('fp16', 'bf16'):
            net = nets.create_model(arch + '_rmac', pretrained='')
            net.load_state_dict(sd)
            net.compute_dtype = dtype
            net.cuda().eval()
            assert net.overflowed() is False            # nothing ran yet
            d = net(x)
            if dtype == 'bf16':
                assert torch.isfinite(d).all() and net.overflowed() is False
                assert torch.isfinite(td.extract_image_features(db, '', net, threads=0)).all()
                continue
            hidden = bool(torch.isfinite(d).all())
            print('\n[overflow] %s fp16: descriptors finite = %s' % (arch, hidden))
            assert net.overflowed() is True, 'an fp16 inf was stored in layer2 and nobody noticed'
            assert net.overflowed() is False            # read-and-clear
            with pytest.raises(FloatingPointError) as ei:
                td.extract_image_features(db, '', net, threads=0)
            assert 'DIRTORCH_AMD_DTYPE=bf16' in str(ei.value)
            if hidden:
                assert 'fp16 overflow inside the trunk' in str(ei.value)
    # weights that do not fit fp16 are refused when the engine is built (DIR_ERR_RANGE): an inf weight would be
    # born on the host, where no kernel's overflow word can see it
    # (fp16p packs the same fp16 hi planes - and a lo plane fp16(w - inf) = -inf next to an inf hi plane would make NaN: a
    # paired layer, layer1's 1x1 conv3, and a single-plane one are both refused)
    for dtype, key, layer in (('fp16', 'layer3.1.bn2.weight', 'layer3.1.conv2'), ('fp16p', 'layer3.1.bn2.weight', 'layer3.1.conv2'),
                              ('fp16p', 'layer1.1.bn3.weight', 'layer1.1.conv3'), ('fp16p', 'bn1.weight', 'conv1')):
        sd = O.synth_state_dict('resnet50', seed=7, gemp=3.0)
        sd[key] = sd[key] * 1e7
        net = nets.create_model('resnet50_rmac', pretrained='')
        net.load_state_dict(sd)
        net.compute_dtype = dtype
        net.cuda().eval()
        with pytest.raises(FloatingPointError) as ei:
            net(x)
        assert layer in str(ei.value) and 'DIRTORCH_AMD_DTYPE=bf16' in str(ei.value), (dtype, key, str(ei.value))
    # bf16 has the range - and says, once per process, what it cannot promise
    from dirtorch_amd.nets import rmac_resnet
    del rmac_resnet._BF16_WARNED[:]
    net.compute_dtype = 'bf16'
    with pytest.warns(RuntimeWarning, match='does not meet the 1e-4'):
        assert torch.isfinite(net(x)).all()
    # a healthy checkpoint never trips the word (fp16, every kernel family of a 1024^2 batch)
    sd = O.synth_state_dict('resnet50', seed=7)
    net = nets.create_model('resnet50_rmac', pretrained='')
    net.load_state_dict(sd)
    net.compute_dtype = 'fp16'
    net.cuda().eval()
    g = torch.Generator(device='cuda').manual_seed(3)
    net(torch.randint(0, 256, (8, 1024, 1024, 3), generator=g, dtype=torch.uint8, device='cuda'))
    assert net.overflowed() is False
