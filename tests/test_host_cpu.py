"""Host-side logic that needs no GPU: datasets + AP protocol, transform DSL, shard arithmetic and
the world_size-2 all-gather path over gloo (the N > 1 code path of bench.py / test_dir.py)."""
import os
import pickle

import numpy as np
import pytest
import torch


# ---- datasets -----------------------------------------------------------------------------------
def write_gnd(tmp_path, gnd, N):
    gt = {'imlist': ['im%04d' % i for i in range(N)], 'qimlist': ['q%d' % i for i in range(len(gnd))],
          'gnd': gnd}
    f = os.path.join(str(tmp_path), 'gnd_synth.pkl')
    with open(f, 'wb') as fh:
        pickle.dump(gt, fh)
    return f


def test_eval_query_ap_matches_reference(tmp_path, postproc_goldens):
    """ImageListRelevants.eval_query_AP vs the reference's own class run on the same pickle
    (tests/golden/make_golden.py, dirtorch/datasets/generic.py:196-224)."""
    from dirtorch_amd import datasets
    g = postproc_goldens
    gnd = pickle.loads(g['evalap.gnd'][0])
    scores = g['evalap.scores']
    db = datasets.ImageListRelevants(write_gnd(tmp_path, gnd, scores.shape[1]), root=str(tmp_path))
    assert len(db) == 200 and db.nquery == 6 and db.get_key(3) == 'im0003.jpg'
    qdb = db.get_query_db()
    assert len(qdb) == 6 and qdb.get_roi(0) == (0, 0, 10, 10)
    for q in range(db.nquery):
        d = db.eval_query_AP(q, scores[q])
        for mode in ('easy', 'medium', 'hard'):
            assert d[mode] == pytest.approx(g['evalap.' + mode][q], abs=1e-12)
    with pytest.raises(AssertionError):
        db.eval_query_AP(0, scores[0][:-1])


def test_classic_protocol_and_known_aps(tmp_path):
    from dirtorch_amd import datasets
    assert datasets.compute_average_precision([0, 1, 2]) == 1.0
    assert datasets.compute_average_precision([1, 3]) == pytest.approx(1 / 3)
    assert datasets.compute_average_precision([2, 5, 9]) == pytest.approx(0.2314814814814815)
    assert datasets.compute_average_precision([]) == 0.0
    gnd = [{'bbx': [0, 0, 1, 1], 'ok': [1, 3], 'junk': [0]}]
    db = datasets.ImageListRelevants(write_gnd(tmp_path, gnd, 5), root=str(tmp_path))
    # ranking by score: 4, 3, 2, 1 (0 is junk) -> positives at ranks 1 and 3
    ap = db.eval_query_AP(0, np.array([9., 1., 2., 3., 4.]))
    assert isinstance(ap, float) and ap == pytest.approx(datasets.compute_average_precision([1, 3]))


def test_dataset_factory(tmp_path, monkeypatch):
    from dirtorch_amd import datasets
    lst = tmp_path / 'list.txt'
    lst.write_text('a.png\nb.png\n\n')
    db = datasets.create('ImageList("%s")' % lst)
    assert len(db) == 2 and db.get_key(1) == 'b.png'
    db = datasets.create('ImageList("%s", root="/data")' % lst)
    assert db.get_filename(0) == '/data/a.png'
    with pytest.raises(NotImplementedError):
        db.get_query_db()
    with pytest.raises(NameError):
        datasets.create('Landmarks_dirty')
    with pytest.raises(SyntaxError):
        datasets.create('ImageList(__import__("os").getcwd())')     # literals only, no eval
    monkeypatch.delenv('DB_ROOT', raising=False)
    with pytest.raises(KeyError):
        datasets.create('ROxford5K')
    monkeypatch.setenv('DB_ROOT', str(tmp_path))
    with pytest.raises(FileNotFoundError):
        datasets.create('RParis6K')


# ---- transforms -----------------------------------------------------------------------------------
def test_transform_dsl():
    from PIL import Image
    from dirtorch_amd.utils import transforms as T
    img = Image.fromarray((np.arange(60 * 80 * 3) % 251).astype(np.uint8).reshape(60, 80, 3))   # w=80, h=60
    pre = dict(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225], input_size=224)
    t = T.create('', to_tensor=True, **pre)(img)
    assert t.shape == (3, 60, 80) and t.dtype == torch.float32
    ref = (torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float() / 255
           - torch.tensor(pre['mean']).view(3, 1, 1)) / torch.tensor(pre['std']).view(3, 1, 1)
    assert torch.equal(t, ref)
    u = T.create('', to_tensor='uint8', **pre)(img)
    assert u.shape == (60, 80, 3) and u.dtype == torch.uint8
    # Scale: smallest side -> size, int(0.5 + ...) rounding (transforms.py:147-170)
    assert T.Scale(256).get_params((640, 480)) == (341, 256)
    assert T.Scale(256, largest=True).get_params((640, 480)) == (256, 192)
    assert T.Scale(0.5).get_params((641, 481)) == (321, 241)
    assert T.Scale((32, 16)).get_params((640, 480)) == (32, 16)
    assert T.create('Scale(30)', to_tensor='uint8', **pre)(img).shape == (30, 40, 3)
    assert T.create('Scale(30), CenterCrop(24)', to_tensor='uint8', **pre)(img).shape == (24, 24, 3)
    assert T.create('Pad(100)', to_tensor='uint8', **pre)(img).shape == (100, 80, 3)
    assert T.create('PadSquare()', to_tensor='uint8', **pre)(img).shape == (80, 80, 3)
    assert T.create('Scale(input_size)', to_tensor=True, **pre)(img).shape == (3, 224, 299)
    with pytest.raises(SyntaxError):
        T.create('RandomCrop(8)', **pre)
    with pytest.raises(SyntaxError):
        T.create('__import__("os").system("true")', **pre)
    # chains the reference's eval() accepts: PIL filter constants and simple arithmetic
    sc = T.create('Scale(2*16, interpolation=Image.BICUBIC)', to_tensor='uint8', **pre)
    assert sc.transforms[0].interpolation == Image.BICUBIC and sc(img).shape == (32, 43, 3)
    assert T.create('Scale(input_size//2)', to_tensor='uint8', **pre)(img).shape == (112, 149, 3)
    with pytest.raises(SyntaxError):       # an unsupported argument is the same SyntaxError, not a bare ValueError
        T.create('Scale(open("/etc/passwd"))', **pre)
    with pytest.raises(SyntaxError):
        T.create('Scale(Image.__class__)', **pre)


def test_small_helpers_follow_the_reference(tmp_path):
    """mkdir's isfile='auto' (convenient.py:11-23) and delete_fc leaving the caller's dict alone
    (nets/__init__.py:67-95: the reference deletes from a local copy)."""
    import synth
    from dirtorch_amd import nets
    from dirtorch_amd.utils.convenient import mkdir
    mkdir(str(tmp_path / 'a' / 'b' / 'out.json'))            # has an extension: a file path
    assert (tmp_path / 'a' / 'b').is_dir() and not (tmp_path / 'a' / 'b' / 'out.json').exists()
    mkdir(str(tmp_path / 'c' / 'd'))                         # no extension: a directory
    assert (tmp_path / 'c' / 'd').is_dir()
    sd = synth.synth_state_dict('resnet18', seed=1, out_dim=64)
    net = nets.create_model('resnet18_rmac', pretrained='', out_dim=64)
    keys = list(sd.keys())
    nets.load_pretrained_weights(net, sd, delete_fc=True)
    assert list(sd.keys()) == keys and 'fc.weight' in sd


def test_loader_batches_even_single_threaded(tmp_path):
    from PIL import Image
    from dirtorch_amd import datasets
    from dirtorch_amd.utils.pytorch_loader import get_loader
    names = []
    for i in range(3):
        Image.fromarray(np.full((20, 30, 3), 40 * i, np.uint8)).save(str(tmp_path / ('i%d.png' % i)))
        names.append('i%d.png' % i)
    db = datasets.ImageList(imgs=names, root=str(tmp_path))
    pre = dict(mean=[0.5] * 3, std=[0.25] * 3, input_size=224)
    batches = list(get_loader(db, '', False, preprocess=pre, output=['img'], batch_size=1, threads=1, shuffle=False))
    assert len(batches) == 3 and batches[1][0].shape == (1, 20, 30, 3) and int(batches[2][0][0, 0, 0, 0]) == 80
    fl = list(get_loader(db, '', False, preprocess=pre, output=['img'], batch_size=3, threads=0, shuffle=False,
                         device_normalize=False))
    assert fl[0][0].shape == (3, 3, 20, 30) and fl[0][0].dtype == torch.float32


# ---- sharding + all-gather over gloo, world size 2 ------------------------------------------------
def test_shard_ranges_cover_everything():
    from dirtorch_amd.distributed import shard_range
    for n in (0, 1, 2, 7, 70, 4993, 1006322):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def _fake_descs(lo, hi, D=16):
    i = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1)
    return torch.sin(i * 0.37 + torch.arange(D, dtype=torch.float32) * 1.3) * (1 + i / 7)


class _FakeNet(object):
    out_dim, iscuda, without_fc = 16, False, False


class _FakeDB(object):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def get_key(self, i):
        return i


def _fake_extract(dataset, trfs, net, **kw):
    keys = [dataset.get_key(i) for i in range(len(dataset))]
    return _fake_descs(keys[0], keys[-1] + 1) if keys else torch.empty(0, 16)


def _worker(rank, world, port, ns, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from dirtorch_amd import distributed as dd
    r, w, _ = dd.init_from_env('gloo')
    assert (r, w) == (rank, world)
    res = {}
    for n in ns:
        lo, hi = dd.shard_range(n)
        res[('gather', n)] = dd.allgather_rows(_fake_descs(lo, hi), n)
        res[('mesh', n)] = dd.allgather_rows(_fake_descs(lo, hi), n, algo='mesh')   # W - 1 direct sends, one group
        res[('extract', n)] = dd.extract_sharded(_fake_extract, _FakeDB(n), '', _FakeNet())
    # a failure on ONE rank's shard (fp16 overflow, test_dir._check_finite) must surface on EVERY rank before
    # the collective, not leave the healthy ranks blocked in it
    def _overflowing_extract(dataset, trfs, net, **kw):
        if dataset.get_key(0) == 0:          # rank 0's shard
            raise FloatingPointError('fp16 overflow inside the trunk with compute dtype fp16')
        return _fake_extract(dataset, trfs, net, **kw)
    try:
        dd.extract_sharded(_overflowing_extract, _FakeDB(10), '', _FakeNet())
        res['overflow'] = 'no error'
    except FloatingPointError as e:
        res['overflow'] = str(e)
    out[rank] = res
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_allgather_equals_single_process_concat_gloo_ws2():
    import torch.multiprocessing as mp
    ns = [1, 2, 7, 64, 4993]
    port = 29000 + (os.getpid() % 2000)
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(2, port, ns, out), nprocs=2, join=True)
        res = dict(out)
    for n in ns:
        full = _fake_descs(0, n)
        for r in (0, 1):
            assert torch.equal(res[r][('gather', n)], full), (n, r)       # bit-for-bit
            assert torch.equal(res[r][('mesh', n)], full), (n, r)         # the full-mesh exchange: the same rows
            assert torch.equal(res[r][('extract', n)], full), (n, r)
    assert 'fp16 overflow inside the trunk' in res[0]['overflow'] and 'another rank' in res[1]['overflow'], \
        (res[0]['overflow'], res[1]['overflow'])


@pytest.mark.timeout(300)
def test_bench_gpus_n_launches_itself_without_torchrun():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment becomes its own launcher (N ranks, rendezvous on
    127.0.0.1, rank 0 prints the one line); --dry-launch runs that plumbing and the path's one exchange step over gloo on
    the CPU.  Without GPUs the real run refuses with the counts, not with a usage error."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    for extra in ([], ['--workload', 'distractors']):
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-launch'] + extra, env=env,
                           capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
        assert line['ranks'] == 2 and line['n_gpus'] == 2 and line['exchange_ok'] is True
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], env=env, capture_output=True,
                           text=True, timeout=240)
        assert r.returncode != 0 and 'needs 2 GPUs, found' in r.stderr, (r.returncode, r.stderr[-500:])


def test_single_process_passthrough():
    from dirtorch_amd import distributed as dd
    x = _fake_descs(0, 9)
    assert dd.allgather_rows(x, 9) is x
    assert dd.rank() == 0 and dd.world_size() == 1
    assert torch.equal(dd.extract_sharded(_fake_extract, _FakeDB(9), '', _FakeNet()), x)


def test_labelled_datasets_and_metrics_match_reference(label_goldens, tmp_path):
    """ImageListLabels / ImageListLabelsQ class-label AP and top-k, accuracy_topk and the quantized
    AP against outputs of the reference (tests/golden/label_goldens.npz)."""
    import numpy as np
    import torch
    from dirtorch_amd import datasets
    from dirtorch_amd.utils import evaluation
    g = label_goldens
    labels, qlabels, scores = list(g['labels.db']), list(g['labels.q']), g['labels.scores']
    (tmp_path / 'db.txt').write_text('\n'.join('im%02d.jpg %s' % (i, l) for i, l in enumerate(labels)) + '\n')
    (tmp_path / 'q.txt').write_text('\n'.join('q%02d.jpg %s' % (i, l) for i, l in enumerate(qlabels)) + '\n')
    cases = (('self', datasets.create('ImageListLabels("%s", root="%s")' % (tmp_path / 'db.txt', tmp_path))),
             ('q', datasets.create('ImageListLabelsQ("%s", "%s", root="%s")' % (tmp_path / 'db.txt', tmp_path / 'q.txt', tmp_path))))
    for tag, db in cases:
        assert db.has_label() and list(g['labels.%s.classes' % tag]) == db.classes
        qdb = db.get_query_db()
        assert (qdb is db) == (tag == 'self')
        aps = np.array([db.eval_query_AP(q, scores[q]) for q in range(qdb.nimg)], dtype=np.float64)
        np.testing.assert_allclose(aps, g['labels.%s.ap' % tag], rtol=0, atol=1e-12)
        keys = list(g['labels.%s.topk_keys' % tag])
        tops = [db.eval_query_top(q, scores[q]) for q in range(qdb.nimg)]
        assert sorted(tops[0]) == keys
        assert np.array_equal(np.array([[t[k] for k in keys] for t in tops]), g['labels.%s.topk' % tag])
    assert -1 in g['labels.q.ap']                       # the query of a class without database images
    with pytest.raises(NotImplementedError):            # unlabelled lists still have no AP
        datasets.ImageList(imgs=['a.jpg']).eval_query_AP(0, np.zeros(1))
    logits, target = g['acc.logits'], g['acc.target']
    np.testing.assert_allclose(evaluation.accuracy_topk(logits, target, topk=(1, 3, 5)), g['acc.np'], atol=1e-12)
    got = evaluation.accuracy_topk(torch.from_numpy(logits), torch.from_numpy(target), topk=(1, 3, 5))
    np.testing.assert_allclose([float(v) for v in got], g['acc.torch'], atol=1e-7)
    np.testing.assert_allclose(evaluation.compute_average_precision_quantized(g['apq.labels'], g['apq.order']),
                               g['apq.value'], atol=1e-7)
    assert evaluation.compute_average_precision_quantized(np.zeros(5, dtype=int), np.arange(5)) == 0


def test_named_list_datasets_resolve_under_db_root(tmp_path, monkeypatch):
    """The Landmarks* names of datasets/landmarks.py and landmarks18.py: fixed list files under DB_ROOT."""
    from dirtorch_amd import datasets
    names = {'Landmarks_clean', 'Landmarks_clean_val', 'Landmarks_lite', 'Landmarks18_train', 'Landmarks18',
             'Landmarks18_lite', 'Landmarks18_mid', 'Landmarks18_5K', 'Landmarks18_val', 'Landmarks18_valdstr',
             'Landmarks18_index', 'Landmarks18_new_index', 'Landmarks18_test', 'Landmarks18_pca',
             'Landmarks18_missing_index'}
    assert names <= set(datasets._REGISTRY)
    monkeypatch.setenv('DB_ROOT', str(tmp_path))
    (tmp_path / 'landmarks18' / 'lists').mkdir(parents=True)
    (tmp_path / 'landmarks18' / 'lists' / 'index.txt').write_text('a/1.jpg\nb/2.jpg\n')
    (tmp_path / 'landmarks' / 'annotations').mkdir(parents=True)
    (tmp_path / 'landmarks' / 'annotations' / 'annotation_clean_val.txt').write_text('x.jpg 7\ny.jpg 7\nz.jpg 9\n')
    idx = datasets.create('Landmarks18_index')
    assert len(idx) == 2 and idx.get_filename(1) == str(tmp_path / 'landmarks18') + '/b/2.jpg' and not idx.has_label()
    val = datasets.create('Landmarks_clean_val')
    assert len(val) == 3 and val.nclass == 2 and val.get_label(2) == '9' and val.get_label(2, toint=True) == 1
    assert val.eval_query_AP(0, np.array([0.1, 0.9, 0.5], dtype=np.float32)) == 1.0
    monkeypatch.delenv('DB_ROOT')
    with pytest.raises(KeyError):
        datasets.create('Landmarks18_pca')


def _rank_counts_model(scores, probe):
    """The arithmetic of csrc/ranking.hip (rank_sort / rank_hist / rank_finalize kernels), restated in NumPy: 64-bit keys
    (order-preserving image of the score, index), probes sorted by key, one lower bound per score, a histogram, suffix
    sums scattered back to the probes' slots."""
    def key(s, idx):
        s = np.where(s == 0, np.float32(0), s).astype(np.float32)          # -0 -> +0
        b = s.view(np.uint32).astype(np.uint64)
        u = np.where(b & 0x80000000, ~b & 0xffffffff, b | 0x80000000)
        return (u << np.uint64(32)) | idx.astype(np.uint64)
    Q, P = probe.shape
    N = scores.shape[1]
    counts = np.zeros((Q, P), np.int64)
    KMAX = np.uint64(0xffffffffffffffff)
    for q in range(Q):
        valid = probe[q] >= 0
        ps = np.where(valid, scores[q, np.maximum(probe[q], 0)], np.float32(np.nan))
        pk = np.where(valid & ~np.isnan(ps), key(np.nan_to_num(ps, nan=0.0), np.maximum(probe[q], 0)), KMAX)
        order = np.lexsort((np.arange(P), pk))                                # ties between equal keys break on the slot
        sk = pk[order]
        ok = ~np.isnan(scores[q])                                             # NaN scores rank before nothing
        kj = key(np.nan_to_num(scores[q][ok], nan=0.0), np.arange(N)[ok])
        k = np.searchsorted(sk, kj, side='left')                              # probes with key < key_j
        hist = np.bincount(k, minlength=P + 1)
        suffix = np.cumsum(hist[::-1])[::-1]                                  # suffix[i] = sum_{k >= i} hist[k]
        counts[q, order] = np.where(pk[order] == KMAX, 0, suffix[1:])         # count of sorted position i = sum_{k > i}
    return counts


def test_sorted_probe_rank_counts_equal_the_definition():
    """The device ranking counts, for every listed image, the items that rank before it under np.argsort(scores)[::-1]
    (score greater, or equal with a larger index).  ranking.hip does it with sorted probes + a binary search per score +
    a histogram; this pins that ARITHMETIC (restated in NumPy above) to the definition on the cases the float compare
    makes delicate: heavy ties, +-0, +-inf, NaN scores and NaN probes, a probe listed twice, unused slots."""
    r = np.random.RandomState(0)
    Q, N, P = 4, 3001, 41
    scores = (r.randint(-4, 5, size=(Q, N)) / 2.0).astype(np.float32)
    scores[1] = r.standard_normal(N).astype(np.float32)
    scores[2, ::5] = 0.0
    scores[2, 1::5] = -0.0
    scores[3, [3, 77]] = np.nan
    scores[3, [4, 500]] = [np.inf, -np.inf]
    probe = np.stack([r.choice(N, P, replace=False) for _ in range(Q)]).astype(np.int32)
    probe[3, :6] = [3, 77, 4, 500, 4, 2999]
    probe[0, ::9] = -1
    want = np.zeros((Q, P), np.int64)
    j = np.arange(N)
    for q in range(Q):
        for k in range(P):
            p = probe[q, k]
            if p >= 0:
                with np.errstate(invalid='ignore'):
                    want[q, k] = np.count_nonzero((scores[q] > scores[q, p]) | ((scores[q] == scores[q, p]) & (j > p)))
    assert (_rank_counts_model(scores, probe) == want).all()
    # and the definition is the reference's order: position in np.argsort(...)[::-1] with ties by descending index
    order = np.lexsort((np.arange(N), scores[1]))[::-1]
    rank = np.empty(N, np.int64)
    rank[order] = np.arange(N)
    assert (want[1] == rank[probe[1]]).all()


def test_unit_range_check_of_the_pair_similarity_kernel():
    """ranking.is_unit_range: what decides between dir_similarity_unit (two fp16 planes of 2^10 x: operands in (-64, 64),
    csrc/sim_split.hip PAIR) and the general six-product kernel.  L2-normalised descriptors qualify; anything at or beyond
    the bound, NaN or inf does not; empty sets do; and the fp16 arithmetic it guards really holds such values: a plane of
    2^10 * 60 is finite in fp16, one of 2^10 * 64 is not."""
    import torch
    from dirtorch_amd import ranking
    g = torch.Generator().manual_seed(3)
    d = torch.nn.functional.normalize(torch.randn(500, 64, generator=g), dim=1)
    assert ranking.is_unit_range(d, d[:7])
    assert ranking.is_unit_range(d * 59.0)
    assert not ranking.is_unit_range(d, d * 1e4)
    bad = d.clone()
    bad[3, 5] = float('nan')
    assert not ranking.is_unit_range(bad)
    bad[3, 5] = float('-inf')
    assert not ranking.is_unit_range(bad)
    bad[3, 5] = -ranking.UNIT_RANGE_BOUND
    assert not ranking.is_unit_range(bad)
    assert ranking.is_unit_range(torch.zeros(0, 64), d)
    assert torch.isfinite(torch.tensor(1024.0 * ranking.UNIT_RANGE_BOUND).half())
    assert not torch.isfinite(torch.tensor(1024.0 * 64.0).half())
    # the pair representation of a unit-vector entry: hi + lo of 2^10 x holds x to ~2^-22
    x = d.flatten()[:4096].double()
    hi = (x * 1024).float().half()
    lo = ((x * 1024).float() - hi.float()).half()
    rec = (hi.double() + lo.double()) / 1024
    assert float((rec - x).abs().max()) < 2.0 ** -22 * float(x.abs().max()) + 2.0 ** -35


def test_pair_similarity_arithmetic_restated_in_numpy():
    """The arithmetic of csrc/sim_split.hip's PAIR form, restated: x * 2^10 = h + l (fp16 planes), score = (h.h' + h.l' + l.h')
    * 2^-20 with exact plane products (11 x 11 bits) summed in wider precision.  On unit vectors that is an fp32-class dot
    product (what the GPU test measures against fp64: 2.5e-7 incl. the fp32 accumulation of the matrix cores); one plane alone
    is fp16-class; and the dropped l.l' term is below 2^-22 of the sum of |products|."""
    r = np.random.RandomState(4)
    Q, N, D = 16, 300, 2048
    q = r.standard_normal((Q, D))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    d = r.standard_normal((N, D))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    q, d = q.astype(np.float32), d.astype(np.float32)
    d[5] = q[0]

    def planes(x):
        s = (x * np.float32(1024.0)).astype(np.float32)      # exact: a power of two
        h = s.astype(np.float16)
        lo = (s - h.astype(np.float32)).astype(np.float16)    # s - h is exact in fp32
        return h.astype(np.float64), lo.astype(np.float64)

    qh, ql = planes(q)
    dh, dl = planes(d)
    ref = q.astype(np.float64) @ d.astype(np.float64).T
    pair = (qh @ dh.T + (qh @ dl.T + ql @ dh.T)) / 2.0 ** 20
    one = (qh @ dh.T) / 2.0 ** 20
    e_pair, e_one = np.abs(pair - ref).max(), np.abs(one - ref).max()
    assert e_pair < 1e-7, e_pair                     # representation + dropped term only (largest on the score of 1: l.l adds up)
    assert e_one > 100 * e_pair                      # a single plane is three decimal digits worse
    assert abs(pair[0, 5] - 1.0) < 1e-7
    dropped = np.abs(ql @ dl.T) / 2.0 ** 20
    assert dropped.max() < 2.0 ** -22 * (np.abs(q).astype(np.float64) @ np.abs(d).astype(np.float64).T).max()


def test_unit_range_verdict_lives_on_the_tensor_object(monkeypatch):
    """ranking.database_is_unit_range remembers its verdict ON the caller's tensor object (round-5 advice: a cache keyed by
    (data_ptr, shape, dtype, version) can be served to a different upload that the allocator placed at the same address):
    one full check per object, a re-check after an in-place torch update, none inherited by another object."""
    from dirtorch_amd import ranking
    calls = []
    monkeypatch.setattr(ranking, 'is_unit_range', lambda *t: (calls.append(1), True)[1])
    a = torch.zeros(4, 4)
    assert ranking.database_is_unit_range(a) and ranking.database_is_unit_range(a)
    assert len(calls) == 1
    a.add_(1)                                   # torch bumps the version counter: looked at again
    assert ranking.database_is_unit_range(a) and len(calls) == 2
    b = torch.zeros(4, 4)                       # another object - wherever it lives - starts without a verdict
    assert ranking.database_is_unit_range(b) and len(calls) == 3
    ranking.forget_unit_range(b)                # contents rewritten through raw pointers: the caller says so
    assert ranking.database_is_unit_range(b) and len(calls) == 4
    assert not hasattr(ranking, '_RANGE_CACHE')
