"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run in the build container only (the GPU box has no /root/reference):

    PYTHONPATH=/root/reference DB_ROOT=/tmp python tests/golden/make_golden.py [model] [head] [postproc] [label] [qe]

What is pinned (reference file:line):
    descriptors      dirtorch.nets.create_model(...)(x)      nets/__init__.py:24, rmac_resnet.py:39-69
    trunk features   ResNet.forward                          backbones/resnet.py:157-174
    FPN / classifier create_model('*_fpn_rmac' | 'resnetNN')(x)  rmac_resnet_fpn.py:50-86, resnet.py:157-174
    pool             dirtorch.utils.common.pool              utils/common.py:41-55
    whiten_features  dirtorch.utils.common.whiten_features   utils/common.py:221-239  (sklearn PCA)
    matmul           dirtorch.utils.common.matmul            utils/common.py:30-38
    AP               compute_average_precision               utils/evaluation.py:46-82
    eval_query_AP    ImageListRelevants.eval_query_AP        datasets/generic.py:196-224
    label AP / top-k Dataset.eval_query_AP / eval_query_top  datasets/dataset.py:71-105 (ImageListLabels[Q])
    accuracy_topk, compute_average_precision_quantized       utils/evaluation.py:8-38,85-98

Weights, images and descriptors come from the deterministic generators of tests/synth.py
(synth_state_dict, synth_images, synth_descriptors; re-exported by oracle/dir_oracle.py), so the fixtures hold only the reference's OUTPUTS (a few hundred KB).
"""
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('DB_ROOT', tempfile.gettempdir())
sys.path.insert(0, '/root/reference')

import dir_oracle as O  # noqa: E402

import dirtorch.nets as ref_nets  # noqa: E402
from dirtorch.utils import common as ref_common  # noqa: E402
from dirtorch.utils.evaluation import compute_average_precision as ref_ap  # noqa: E402
from dirtorch.datasets.generic import ImageListRelevants  # noqa: E402

torch.set_num_threads(8)

# (tag, arch, model options, gemp in checkpoint, B, H, W)
MODEL_CASES = [
    ('r18_gem', 'resnet18', dict(), 2.7, 2, 96, 80),
    ('r50_gem', 'resnet50', dict(), 2.7, 2, 97, 75),          # odd H and W
    ('r50_gem_b1', 'resnet50', dict(), 3.0, 1, 64, 64),       # B == 1 -> shape (D,)
    ('r50_max_norm', 'resnet50', dict(pooling='max', norm_features=True), None, 2, 64, 96),
    ('r50_avg_cb', 'resnet50', dict(pooling='avg', center_bias=0.5, out_dim=512), None, 2, 64, 64),
    ('r50_nofc', 'resnet50', dict(without_fc=True), 2.2, 2, 64, 64),
    ('r101_gem', 'resnet101', dict(), 2.7, 1, 128, 96),
]


def model_goldens():
    out = {}
    for tag, arch, opts, gemp, B, H, W in MODEL_CASES:
        pooling = opts.get('pooling', 'gem')
        sd = O.synth_state_dict(arch, seed=7, out_dim=opts.get('out_dim', 2048),
                                gemp=gemp if gemp else 3.0, pooling=pooling)
        net = ref_nets.create_model(arch + '_rmac', pretrained='', **opts)
        net.load_state_dict(sd)
        net.eval()
        x = O.synth_images(11, B, H, W)
        with torch.no_grad():
            feat = ref_nets.rmac_resnet.ResNet.forward(net, x)
            desc = net(x.clone())
        out[tag + '.desc'] = desc.numpy()
        # a thin slice of the trunk output is enough to pin ResNet.forward (the full map is MBs)
        out[tag + '.feat_slice'] = feat[:, ::64, :, :].numpy()
        out[tag + '.feat_shape'] = np.array(feat.shape)
        print(tag, 'desc', tuple(desc.shape), 'feat', tuple(feat.shape))
    np.savez_compressed(os.path.join(HERE, 'model_goldens.npz'), **out)


# (tag, reference factory name, oracle head, arch, model options, B, H, W)
HEAD_CASES = [
    ('r50_fpn', 'resnet50_fpn_rmac', 'fpn', 'resnet50', dict(), 2, 97, 75),       # 4x3 -> 7x5 upsample
    ('r18_fpn_norm', 'resnet18_fpn_rmac', 'fpn', 'resnet18', dict(norm_features=True), 2, 96, 80),
    ('r50_fpn_b1', 'resnet50_fpn_rmac', 'fpn', 'resnet50', dict(out_dim=512), 1, 64, 64),
    ('r50_fpn_nofc', 'resnet50_fpn_rmac', 'fpn', 'resnet50', dict(without_fc=True), 2, 64, 96),
    ('r101_fpn0', 'resnet101_fpn0_rmac', 'fpn0', 'resnet101', dict(), 1, 128, 96),
    ('r50_cls', 'resnet50', 'cls', 'resnet50', dict(out_dim=1000), 2, 64, 80),
    ('r18_cls_b1', 'resnet18', 'cls', 'resnet18', dict(out_dim=256), 1, 64, 64),
]


def head_goldens():
    """The FPN heads (rmac_resnet_fpn.py:50-86) and the plain classifier (resnet.py:157-174)."""
    out = {}
    for tag, factory, head, arch, opts, B, H, W in HEAD_CASES:
        feat = 512 * (4 if O.ARCH[arch][0] else 1)
        default_out = feat + feat // 2 if head in ('fpn', 'fpn0') else 2048
        sd = O.synth_state_dict(arch, seed=9, out_dim=opts.get('out_dim', default_out), gemp=2.6,
                                pooling='gem', head=head)
        net = ref_nets.create_model(factory, pretrained='', **opts)
        net.load_state_dict(sd)
        net.eval()
        x = O.synth_images(13, B, H, W)
        with torch.no_grad():
            desc = net(x.clone())
        out[tag + '.desc'] = desc.numpy()
        print(tag, 'out', tuple(desc.shape))
    np.savez_compressed(os.path.join(HERE, 'head_goldens.npz'), **out)


def label_goldens():
    """Class-label evaluation (datasets/dataset.py:71-105, generic.py:44-121, utils/evaluation.py)."""
    np.bool8 = np.bool_     # shim: dataset.py:99 uses an alias NumPy 2 removed
    from dirtorch.datasets import generic as ref_generic
    from dirtorch.utils import evaluation as ref_eval
    r = np.random.RandomState(5)
    labels = ['c%d' % r.randint(0, 6) for _ in range(40)]
    qlabels = ['c%d' % r.randint(0, 7) for _ in range(9)]       # 'c6' has no database image
    scores = r.standard_normal((40, 40)).astype(np.float32)
    out = {'labels.db': np.array(labels), 'labels.q': np.array(qlabels), 'labels.scores': scores}
    with tempfile.TemporaryDirectory() as d:
        open(d + '/db.txt', 'w').write('\n'.join('im%02d.jpg %s' % (i, l) for i, l in enumerate(labels)) + '\n')
        open(d + '/q.txt', 'w').write('\n'.join('q%02d.jpg %s' % (i, l) for i, l in enumerate(qlabels)) + '\n')
        for tag, db in (('self', ref_generic.ImageListLabels(d + '/db.txt', root=d)),
                        ('q', ref_generic.ImageListLabelsQ(d + '/db.txt', d + '/q.txt', root=d))):
            nq = db.get_query_db().nimg
            out['labels.%s.classes' % tag] = np.array(db.classes)
            out['labels.%s.ap' % tag] = np.array([db.eval_query_AP(q, scores[q]) for q in range(nq)], dtype=np.float64)
            tops = [db.eval_query_top(q, scores[q]) for q in range(nq)]
            out['labels.%s.topk' % tag] = np.array([[t[k] for k in sorted(t)] for t in tops])
            out['labels.%s.topk_keys' % tag] = np.array(sorted(tops[0]))
    logits = r.standard_normal((12, 7)).astype(np.float32)
    target = r.randint(0, 7, 12)
    out['acc.logits'], out['acc.target'] = logits, target
    out['acc.np'] = np.array(ref_eval.accuracy_topk(logits, target, topk=(1, 3, 5)), dtype=np.float64)
    out['acc.torch'] = np.array([float(v) for v in ref_eval.accuracy_topk(
        torch.from_numpy(logits), torch.from_numpy(target), topk=(1, 3, 5))])
    rel = (r.uniform(size=50) < 0.2).astype(np.int64)
    order = np.argsort(-r.standard_normal(50))
    out['apq.labels'], out['apq.order'] = rel, order
    out['apq.value'] = np.array(ref_eval.compute_average_precision_quantized(rel, order))
    np.savez_compressed(os.path.join(HERE, 'label_goldens.npz'), **out)
    print('label goldens written')


def postproc_goldens():
    from sklearn.decomposition import PCA
    r = np.random.RandomState(3)
    out = {}
    # multi-scale pooling
    xs = [torch.from_numpy(r.standard_normal((5, 64)).astype(np.float32)) for _ in range(3)]
    xs[1][0, :4] = 0.0  # exercises sign(0) = 0
    out['pool.in'] = torch.stack(xs).numpy()
    out['pool.mean'] = ref_common.pool(xs, 'mean').numpy()
    out['pool.gem3'] = ref_common.pool(xs, 'gem', 3).numpy()
    out['pool.gem2.5'] = ref_common.pool(xs, 'gem', 2.5).numpy()
    # PCA whitening with a real sklearn object
    base = r.standard_normal((300, 96)).astype(np.float32) @ r.standard_normal((96, 96)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    pca = PCA(whiten=True)
    pca.fit(base)
    X = base[:40] + 0.01 * r.standard_normal((40, 96)).astype(np.float32)
    out['pca.mean'] = pca.mean_
    out['pca.components'] = pca.components_
    out['pca.var'] = pca.explained_variance_
    out['whiten.in'] = X
    out['whiten.p0.5'] = ref_common.whiten_features(X, pca, whitenp=0.5)
    out['whiten.p0.25_v32_m2'] = ref_common.whiten_features(X, pca, whitenp=0.25, whitenv=32, whitenm=2.0)
    out['whiten.nol2'] = ref_common.whiten_features(X, pca, l2norm=False, whitenp=0.5)
    # similarity
    A = r.standard_normal((7, 96)).astype(np.float32)
    Bm = r.standard_normal((33, 96)).astype(np.float32)
    out['matmul.A'] = A
    out['matmul.B'] = Bm
    out['matmul.np'] = ref_common.matmul(A, Bm)
    out['matmul.torch'] = ref_common.matmul(torch.from_numpy(A), torch.from_numpy(Bm))
    # AP known answers
    ranks = [[0, 1, 2], [1, 3], [0], [], [2, 5, 9], [0, 7, 8, 30]]
    out['ap.values'] = np.array([ref_ap(np.array(k)) for k in ranks])
    out['ap.ranks'] = np.array([','.join(map(str, k)) for k in ranks])
    # revisitop-protocol eval_query_AP on a synthetic ground truth
    N, Q = 200, 6
    gnd = []
    for q in range(Q):
        perm = r.permutation(N)
        gnd.append({'bbx': [0, 0, 10, 10], 'easy': sorted(perm[:5].tolist()),
                    'hard': sorted(perm[5:12].tolist()), 'junk': sorted(perm[12:20].tolist())})
    gnd[4]['easy'] = []   # a query without easy positives -> AP -1 in 'easy' mode
    gt = {'imlist': ['im%04d' % i for i in range(N)], 'qimlist': ['q%d' % i for i in range(Q)], 'gnd': gnd}
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, 'gnd_synth.pkl')
        with open(f, 'wb') as fh:
            pickle.dump(gt, fh)
        db = ImageListRelevants(f, root=d)
    scores = r.standard_normal((Q, N)).astype(np.float32)
    scores[2, 10] = scores[2, 11]  # a tie
    aps = [db.eval_query_AP(q, scores[q]) for q in range(Q)]
    out['evalap.scores'] = scores
    out['evalap.gnd'] = np.array([pickle.dumps(gnd)], dtype=object)
    for mode in ('easy', 'medium', 'hard'):
        out['evalap.' + mode] = np.array([a[mode] for a in aps], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'postproc_goldens.npz'), **out)
    print('postproc goldens written')


# (k, alpha) pairs of the alpha-QE / DBA goldens; mirrored by tests (QE_CASES)
QE_CASES = [(1, 0), (1, 3), (5, 0), (5, 3), (3, 1)]


def qe_goldens():
    """alpha query expansion / database augmentation: expand_descriptors (test_dir.py:24-44), the
    self-set form (--adba: db=None, diagonal zeroed) and the db= form (--aqe)."""
    from dirtorch.test_dir import expand_descriptors as ref_expand
    import synth
    q = synth.synth_descriptors(31, 12, 64)
    db = synth.synth_descriptors(32, 40, 64)
    out = {}
    for k, alpha in QE_CASES:
        out['qe.self.k%d.a%d' % (k, alpha)] = ref_expand(db.copy(), alpha=alpha, k=k)
        out['qe.db.k%d.a%d' % (k, alpha)] = ref_expand(q.copy(), db=db.copy(), alpha=alpha, k=k)
    assert ref_expand(q, db=db, alpha=3, k=0) is q           # k == 0: the input object itself
    np.savez_compressed(os.path.join(HERE, 'qe_goldens.npz'), **out)
    print('qe goldens written', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    which = sys.argv[1:] or ['model', 'head', 'postproc', 'label', 'qe']
    if 'qe' in which:
        qe_goldens()
    if 'model' in which:
        model_goldens()
    if 'head' in which:
        head_goldens()
    if 'postproc' in which:
        postproc_goldens()
    if 'label' in which:
        label_goldens()
