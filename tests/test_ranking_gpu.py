"""Device-side ranking / AP against the host protocol
(ImageListRelevants.eval_query_AP, pinned to the reference in tests/test_host_cpu.py)."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_db(tmp_path, N, Q, r, classic=False, npos=12, njunk=6):
    from dirtorch_amd import datasets
    gnd = []
    for q in range(Q):
        idx = r.choice(N, npos + njunk, replace=False)
        if classic:
            gnd.append({'bbx': [0, 0, 1, 1], 'ok': sorted(idx[:npos].tolist()), 'junk': sorted(idx[npos:].tolist())})
        else:
            gnd.append({'bbx': [0, 0, 1, 1], 'easy': sorted(idx[:npos // 2].tolist()),
                        'hard': sorted(idx[npos // 2:npos].tolist()), 'junk': sorted(idx[npos:].tolist())})
    if not classic:
        gnd[1]['easy'] = []                       # AP -1 in 'easy' mode
        gnd[2]['junk'] = gnd[2]['junk'] + gnd[2]['hard'][:1]   # listed as hard AND junk: junk wins
    f = os.path.join(str(tmp_path), 'gnd.pkl')
    with open(f, 'wb') as fh:
        pickle.dump({'imlist': ['i%d' % i for i in range(N)], 'qimlist': ['q%d' % i for i in range(Q)], 'gnd': gnd}, fh)
    return datasets.ImageListRelevants(f, root=str(tmp_path)), gnd


def test_rank_counts_kernel_matches_argsort():
    from dirtorch_amd import ops
    r = np.random.RandomState(0)
    Q, N, P = 5, 10007, 37                      # N not a multiple of the 4096 chunk
    scores = r.standard_normal((Q, N)).astype(np.float32)
    scores[0, 17] = scores[0, 9000] = scores[0, 4096]       # ties: larger index ranks first
    probe = np.stack([r.choice(N, P, replace=False) for _ in range(Q)]).astype(np.int32)
    probe[0, :3] = [17, 9000, 4096]
    probe[3, 30:] = -1
    c, s = ops.rank_counts(torch.from_numpy(scores).cuda(), torch.from_numpy(probe).cuda())
    c, s = c.cpu().numpy(), s.cpu().numpy()
    for q in range(Q):
        order = np.lexsort((np.arange(N), scores[q]))[::-1]      # score desc, then index desc
        rank = np.empty(N, np.int64)
        rank[order] = np.arange(N)
        for k in range(P):
            if probe[q, k] < 0:
                assert c[q, k] == 0
            else:
                assert c[q, k] == rank[probe[q, k]], (q, k)
                assert s[q, k] == scores[q, probe[q, k]]


@pytest.mark.parametrize('classic', [False, True])
def test_device_ap_equals_host_protocol(tmp_path, classic):
    from dirtorch_amd import ranking
    r = np.random.RandomState(1)
    N, Q = 4993, 9
    db, gnd = make_db(tmp_path, N, Q, r, classic)
    scores = np.stack([r.permutation(N) for _ in range(Q)]).astype(np.float32) / N      # distinct: no ties
    host = [db.eval_query_AP(q, scores[q]) for q in range(Q)]
    dev = ranking.eval_aps_device(db, torch.from_numpy(scores).cuda())
    for h, d in zip(host, dev):
        if classic:
            assert d == pytest.approx(h, abs=1e-12)
        else:
            for m in ('easy', 'medium', 'hard'):
                assert d[m] == pytest.approx(h[m], abs=1e-12), m
    if not classic:
        assert host[1]['easy'] == -1 and dev[1]['easy'] == -1


def test_million_distractors_ranking(tmp_path):
    """BASELINE config D on one GPU at its real dimensions: 70 queries x (6322 + 1e6 distractors) x
    2048-d fp32 descriptors (8.2 GB, generated on the device).  Checks: the similarity kernel against
    fp64 dot products of sampled entries; device AP == host protocol (ImageListRelevants.eval_query_AP
    over the downloaded row) for sampled queries; planted positives are actually retrieved."""
    from dirtorch_amd import ranking
    r = np.random.RandomState(2)
    Nb, Nd, Q, D = 6322, 1000000, 70, 2048
    N = Nb + Nd
    db, gnd = make_db(tmp_path, N, Q, r, npos=40, njunk=10)
    g = torch.Generator(device='cuda').manual_seed(3)
    base = torch.empty(N, D, device='cuda')
    for i in range(0, N, 131072):                        # generated chunk-wise: no 8 GB temporaries
        base[i:i + 131072] = torch.randn(min(131072, N - i), D, generator=g, device='cuda')
    qs = torch.randn(Q, D, generator=g, device='cuda')
    for q in range(Q):                                   # plant the positives near their query
        idx = torch.tensor(gnd[q]['easy'] + gnd[q]['hard'], device='cuda')
        base[idx] += qs[q] * torch.rand(len(idx), 1, generator=g, device='cuda') * 1.5
    for i in range(0, N, 131072):
        base[i:i + 131072] = torch.nn.functional.normalize(base[i:i + 131072], dim=1)
    qs = torch.nn.functional.normalize(qs, dim=1)
    scores = ranking.similarity_device(qs, base)
    assert scores.shape == (Q, N)
    cols = torch.from_numpy(r.choice(N, 512, replace=False)).cuda()
    ref = (qs.double() @ base[cols].double().t()).cpu().numpy()
    got = scores[:, cols].cpu().numpy()
    assert np.abs(got - ref).max() < 2e-6, np.abs(got - ref).max()      # fp32 chain over K = 2048, |s| <= 1
    tables = ranking.build_probe_tables(db)
    dev = ranking.eval_aps_device(db, scores, tables)
    sc = scores.cpu().numpy()
    for q in (0, 1, 2, 35, 69):
        host = db.eval_query_AP(q, sc[q])
        for m in ('easy', 'medium', 'hard'):
            assert dev[q][m] == pytest.approx(host[m], abs=1e-12), (q, m)
    med = np.mean([d['medium'] for d in dev if d['medium'] >= 0])
    assert 0.05 < med <= 1.0


def test_many_probes_per_query(tmp_path):
    """More than 1024 listed images for a query (RParis6K has queries with > 1000 positives + junk):
    the probe list spans several kernel launches; device AP still equals the host protocol."""
    from dirtorch_amd import ranking
    r = np.random.RandomState(4)
    N, Q = 20000, 3
    db, gnd = make_db(tmp_path, N, Q, r, npos=1500, njunk=700)
    # distinct scores (20000 float32 normals collide a few times per row, and np.argsort leaves the order
    # of equal scores among the kept images unspecified), plus one planted tie the protocol does define:
    scores = np.stack([r.permutation(N) for _ in range(Q)]).astype(np.float32) / N
    scores[0, gnd[0]['easy'][0]] = scores[0, gnd[0]['junk'][0]]      # a positive tied with a junk image
    host = [db.eval_query_AP(q, scores[q]) for q in range(Q)]
    dev = ranking.eval_aps_device(db, torch.from_numpy(scores).cuda())
    for h, d in zip(host, dev):
        for m in ('easy', 'medium', 'hard'):
            assert d[m] == pytest.approx(h[m], abs=1e-12), m


# ---- N3: alpha query expansion / database augmentation ------------------------------------------------
from test_oracle_golden import QE_CASES, qe_inputs  # noqa: E402


@pytest.mark.parametrize('k,alpha', QE_CASES)
def test_expand_descriptors_vs_reference_golden(k, alpha, qe_goldens):
    """dirtorch_amd.test_dir.expand_descriptors (top-k + weighted mean + L2 on the GPU) against outputs
    of the reference's expand_descriptors (test_dir.py:24-44): the --adba form (self-set, diagonal
    zeroed) and the --aqe form (db=)."""
    from dirtorch_amd import test_dir as td
    q, db = qe_inputs()
    got_self = td.expand_descriptors(db.copy(), alpha=alpha, k=k)
    got_db = td.expand_descriptors(q.copy(), db=db.copy(), alpha=alpha, k=k)
    assert isinstance(got_self, np.ndarray) and got_self.dtype == np.float32
    np.testing.assert_allclose(got_self, qe_goldens['qe.self.k%d.a%d' % (k, alpha)], rtol=0, atol=2e-6)
    np.testing.assert_allclose(got_db, qe_goldens['qe.db.k%d.a%d' % (k, alpha)], rtol=0, atol=2e-6)
    assert td.expand_descriptors(q, db=db, alpha=alpha, k=0) is q
    assert td.expand_descriptors(torch.from_numpy(q), db=torch.from_numpy(db), alpha=alpha, k=k).shape == q.shape


def test_expand_descriptors_at_dataset_scale_and_errors():
    """ROxford5K-sized DBA (4993 x 2048, k = 10, alpha = 3) and AQE (70 queries) against the oracle; the
    row-chunked path (scratch smaller than the score matrix); argument errors."""
    import dir_oracle as O
    import synth
    from dirtorch_amd import ops
    from dirtorch_amd import test_dir as td
    db = synth.synth_descriptors(41, 4993, 2048, clusters=40)
    q = synth.synth_descriptors(42, 70, 2048, clusters=40)
    ref_q = O.expand_descriptors(q, db=db, alpha=3, k=10)
    got_q = td.expand_descriptors(q, db=db, alpha=3, k=10)
    assert np.all(1 - O.cosine(got_q, ref_q) < 1e-6)
    sub = db[:600]
    ref_s = O.expand_descriptors(sub.copy(), alpha=2, k=7)
    got_s = td.expand_descriptors(sub.copy(), alpha=2, k=7)
    assert np.all(1 - O.cosine(got_s, ref_s) < 1e-6)
    chunked = ops.expand_descriptors(torch.from_numpy(sub).cuda(), None, alpha=2.0, k=7,
                                     scratch_bytes=37 * 600 * 4).cpu().numpy()      # 37 rows per pass
    # another GEMM tiling of the same products (fp32 summation order): equal to rounding, not bit for bit
    np.testing.assert_allclose(chunked, got_s, rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        td.expand_descriptors(q[:5], db=db[:3], alpha=1, k=4)            # k > candidates, as np.argpartition
    with pytest.raises(AssertionError):
        td.expand_descriptors(q, db=db, alpha=-1, k=2)


def test_descriptor_widths_that_are_not_multiples_of_four(postproc_goldens):
    """--whitenv N with N % 4 != 0 (the reference accepts any N, common.py:226): whitening to 50 / 127
    components and the similarity of those descriptors run through the element-wise gather of the
    fp32 GEMM and agree with the oracle."""
    import dir_oracle as O
    from dirtorch_amd.utils import common
    G = postproc_goldens
    pca = O.PCAParams(G['pca.mean'], G['pca.components'], G['pca.var'], True)
    X = G['whiten.in']
    for v in (50, 31, 1):
        ref = O.whiten_features(X, pca, whitenp=0.25, whitenv=v)
        got = common.whiten_features(X, pca, whitenp=0.25, whitenv=v)
        assert got.shape == ref.shape == (X.shape[0], v)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
        s_ref = O.matmul(ref[:7], ref)
        s_got = common.matmul(got[:7], got)
        np.testing.assert_allclose(s_got, s_ref, rtol=0, atol=5e-5)
    r = np.random.RandomState(9)
    A = r.standard_normal((70, 127)).astype(np.float32)
    B = r.standard_normal((300, 127)).astype(np.float32)
    np.testing.assert_allclose(common.matmul(A, B), A.astype(np.float64) @ B.astype(np.float64).T, rtol=0, atol=5e-5)
