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


def _host_counts(scores, probe):
    """counts[q][k] by the definition in include/dir_engine.h: items with a greater score, or an equal score and a
    larger index (float compares: NaN ranks before nothing, nothing ranks before NaN, -0 == +0)."""
    Q, P = probe.shape
    out = np.zeros((Q, P), np.int64)
    j = np.arange(scores.shape[1])
    for q in range(Q):
        for k in range(P):
            p = probe[q, k]
            if p >= 0:
                with np.errstate(invalid='ignore'):
                    out[q, k] = np.count_nonzero((scores[q] > scores[q, p]) | ((scores[q] == scores[q, p]) & (j > p)))
    return out


@pytest.mark.parametrize('P', [1, 2, 200, 1024, 4096, 4097, 9001])
def test_rank_counts_probe_slices_and_heavy_ties(P):
    """Every probe-table size class of the sorted-probe kernels (one probe, powers of two, one past a slice of 4096,
    several slices) on scores with many exact ties (values from a small set), so that the (score, index) order is
    exercised everywhere; unused slots scattered through the rows."""
    from dirtorch_amd import ops
    r = np.random.RandomState(P)
    Q, N = 3, 40003
    scores = (r.randint(-6, 7, size=(Q, N)) / 4.0).astype(np.float32)      # 13 distinct values: ties everywhere
    scores[1] = r.standard_normal(N).astype(np.float32)                    # and one row without
    probe = np.stack([r.choice(N, P, replace=False) for _ in range(Q)]).astype(np.int32)
    probe[2, ::7] = -1
    c, s = ops.rank_counts(torch.from_numpy(scores).cuda(), torch.from_numpy(probe).cuda())
    c, s = c.cpu().numpy(), s.cpu().numpy()
    assert (c == _host_counts(scores, probe)).all()
    valid = probe >= 0
    assert (s[valid] == np.take_along_axis(scores, np.maximum(probe, 0), 1)[valid]).all() and (c[~valid] == 0).all()


def test_rank_counts_special_values():
    """NaN scores rank before nothing and nothing ranks before a NaN probe; -0 ties with +0 (index decides);
    infinities order like floats; a probe listed twice gets the same count in both slots."""
    from dirtorch_amd import ops
    r = np.random.RandomState(3)
    Q, N = 2, 20000
    scores = r.standard_normal((Q, N)).astype(np.float32)
    scores[0, [5, 17, 9000, 19999]] = [0.0, -0.0, 0.0, -0.0]
    scores[0, [6, 7000]] = np.nan
    scores[0, [8, 12000]] = [np.inf, -np.inf]
    scores[1, ::3] = 0.0
    scores[1, 1::3] = -0.0
    probe = np.stack([r.choice(N, 64, replace=False) for _ in range(Q)]).astype(np.int32)
    probe[0, :10] = [5, 17, 9000, 19999, 6, 7000, 8, 12000, 5, 8]          # incl. NaN probes and two duplicates
    c, s = ops.rank_counts(torch.from_numpy(scores).cuda(), torch.from_numpy(probe).cuda())
    c, s = c.cpu().numpy(), s.cpu().numpy()
    assert (c == _host_counts(scores, probe)).all()
    assert c[0, 4] == 0 and c[0, 5] == 0 and np.isnan(s[0, 4]) and np.isnan(s[0, 5])
    assert c[0, 0] == c[0, 8] and c[0, 6] == c[0, 9] == 0                   # +inf: nothing ranks before it


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
    checked = 0
    for q in (0, 1, 2, 35, 69, 3, 4, 5):
        # among 10^6 fp32 scores a listed image now and then ties EXACTLY with a distractor; the reference's
        # np.argsort leaves that order unspecified (the device rule is the stable one), so such a query can
        # differ by one rank swap and is not a parity case
        listed = gnd[q]['easy'] + gnd[q]['hard'] + gnd[q]['junk']
        if any((sc[q] == sc[q][p]).sum() > 1 for p in listed):
            continue
        host = db.eval_query_AP(q, sc[q])
        for m in ('easy', 'medium', 'hard'):
            assert dev[q][m] == pytest.approx(host[m], abs=1e-12), (q, m)
        checked += 1
    assert checked >= 4
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
    # k beyond one pick buffer (256): the reference accepts any k <= m (np.argpartition, test_dir.py:36)
    ref_k = O.expand_descriptors(q[:8], db=db[:900], alpha=1, k=300)
    got_k = td.expand_descriptors(q[:8], db=db[:900], alpha=1, k=300)
    assert np.all(1 - O.cosine(got_k, ref_k) < 1e-6)
    all_k = td.expand_descriptors(q[:3], db=db[:520], alpha=0, k=520)    # k == m: the plain mean of everything
    want = (q[:3] + db[:520].sum(0)) / 521
    assert np.all(1 - O.cosine(all_k, want / np.linalg.norm(want, axis=1, keepdims=True)) < 1e-6)
    # a NaN descriptor (e.g. from --load-feats) propagates as a NaN row, as in the reference's arithmetic; the
    # other rows are untouched and nothing reads out of bounds
    bad = q[:4].copy()
    bad[1, 7] = np.nan
    got_bad = td.expand_descriptors(bad, db=db[:100], alpha=1, k=5)
    assert np.isnan(got_bad[1]).all() and np.isfinite(got_bad[[0, 2, 3]]).all()
    assert np.all(1 - O.cosine(got_bad[[0, 2, 3]], O.expand_descriptors(q[:4], db=db[:100], alpha=1, k=5)[[0, 2, 3]]) < 1e-6)
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


# ---- large-database similarity: three-plane bf16 split on the matrix cores (csrc/sim_split.hip) ---------------
@pytest.mark.parametrize('Q,N,D', [(70, 40000, 2048), (1, 33000, 64), (97, 32768 + 255, 512), (200, 50001, 128)])
def test_split_similarity_vs_fp64(Q, N, D, monkeypatch):
    """dir_similarity on a database long enough to take the split kernel (N >= 32768, D % 32 == 0), against fp64
    dot products of the same fp32 data and against the exact fp32 MFMA chain (DIRTORCH_AMD_SIM_EXACT=1): unit
    vectors, no worse than the exact chain (whose error on a score of 1 is ~1e-6: K roundings of a growing sum), row tails (N % 256), more than one 96-row
    query block, a single query."""
    from dirtorch_amd import ops
    g = torch.Generator(device='cuda').manual_seed(Q * 7 + D)
    db = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device='cuda'), dim=1)
    qs = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device='cuda'), dim=1)
    db[5] = qs[0]                                    # a score of 1
    db[N - 1] = -qs[Q - 1]                           # and one of -1, in the last (partial) tile
    got = ops.similarity(qs, db)
    assert got.shape == (Q, N)
    ref = qs.double() @ db.double().t()
    err = float((got.double() - ref).abs().max())
    monkeypatch.setenv('DIRTORCH_AMD_SIM_EXACT', '1')
    exact = ops.similarity(qs, db)
    err_exact = float((exact.double() - ref).abs().max())
    print('\n[split-similarity] %dx%dx%d: max |split - fp64| %.2e, max |exact chain - fp64| %.2e' % (Q, N, D, err, err_exact))
    assert err < 2e-6, err                                  # the bound of the 10^6-row test above
    assert err <= err_exact + 1e-7, (err, err_exact)        # at least as accurate as the fp32 chain it replaces
    assert float(got[0, 5]) == pytest.approx(1.0, abs=3e-7) and float(got[Q - 1, N - 1]) == pytest.approx(-1.0, abs=3e-7)


@pytest.mark.parametrize('Q,N,D', [(70, 40000, 2048), (1, 33000, 64), (97, 32768 + 255, 512), (200, 50001, 128)])
def test_pair_similarity_vs_fp64(Q, N, D):
    """dir_similarity_unit (csrc/sim_split.hip PAIR: two fp16 planes of 2^10 x, three products) on unit vectors: against
    fp64 it is held to the bound of the six-product kernel, and it must be at least as close as that kernel on the same
    data; scores of exactly +-1; row tails, several query blocks, a single query."""
    from dirtorch_amd import ops
    g = torch.Generator(device='cuda').manual_seed(Q * 7 + D)
    db = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device='cuda'), dim=1)
    qs = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device='cuda'), dim=1)
    db[5] = qs[0]
    db[N - 1] = -qs[Q - 1]
    ref = qs.double() @ db.double().t()
    got = ops.similarity(qs, db, unit_range=True)
    six = ops.similarity(qs, db)
    err, err6 = float((got.double() - ref).abs().max()), float((six.double() - ref).abs().max())
    print('\n[pair-similarity] %dx%dx%d: max |pair - fp64| %.2e, max |six-product - fp64| %.2e' % (Q, N, D, err, err6))
    assert got.shape == (Q, N) and err < 2e-6
    assert err <= err6 + 1e-7, (err, err6)
    assert float(got[0, 5]) == pytest.approx(1.0, abs=3e-7) and float(got[Q - 1, N - 1]) == pytest.approx(-1.0, abs=3e-7)
    assert torch.equal(got, ops.similarity(qs, db, unit_range=True))      # run-to-run identical


def test_pair_similarity_range_and_selection():
    """What the fp16 planes can hold: entries up to 60 in magnitude and down to 1e-6 keep fp32-class products (relative to
    sum |q||d|); a value beyond 64 makes the scores of its row NON-FINITE (never silently wrong); ranking.similarity_device
    looks before it picks the kernel; small databases are the exact chain either way."""
    from dirtorch_amd import ops, ranking
    Q, N, D = 40, 33000, 256
    g = torch.Generator(device='cuda').manual_seed(6)
    db = torch.randn(N, D, generator=g, device='cuda')
    qs = torch.randn(Q, D, generator=g, device='cuda')
    db[::3] *= 10.0                       # |values| up to ~52
    db[1::3] *= 1e-5
    db[7] = 0
    qs[::2] *= 1e-3
    assert ranking.is_unit_range(qs, db)
    got = ops.similarity(qs, db, unit_range=True).double()
    ref = qs.double() @ db.double().t()
    scale = qs.double().abs() @ db.double().abs().t()
    rel = (got - ref).abs() / scale.clamp_min(1e-300)
    rel[:, 7] = 0
    assert torch.isfinite(got).all() and float(got[:, 7].abs().max()) == 0.0
    assert float(rel.max()) < 1e-6, float(rel.max())
    assert torch.equal(ranking.similarity_device(qs, db), ops.similarity(qs, db, unit_range=True))
    db[11, 3] = 100.0                     # outside the range
    bad = ops.similarity(qs, db, unit_range=True)
    assert not torch.isfinite(bad[:, 11]).all() and torch.isfinite(bad[:, :11]).all() and torch.isfinite(bad[:, 12:]).all()
    assert not ranking.is_unit_range(qs, db)
    safe = ranking.similarity_device(qs, db)              # ... so the default path takes the six-product kernel
    assert torch.isfinite(safe).all() and torch.equal(safe, ops.similarity(qs, db))
    small = db[:1000].contiguous()
    assert torch.equal(ops.similarity(qs, small, unit_range=True), ops.similarity(qs, small))


def test_split_similarity_keeps_the_fp32_exponent_range(monkeypatch):
    """Not only unit vectors: rows scaled by 2^+-40 (bf16 planes keep all 8 exponent bits, so no scaling step
    exists to go wrong), an all-zero row, and a database row pitch view.  Relative to sum_k |q_k||d_k| the
    error stays at the 1e-7 level."""
    from dirtorch_amd import ops
    Q, N, D = 40, 33000, 256
    g = torch.Generator(device='cuda').manual_seed(5)
    db = torch.randn(N, D, generator=g, device='cuda')
    qs = torch.randn(Q, D, generator=g, device='cuda')
    db[::3] *= 2.0 ** 40
    db[1::3] *= 2.0 ** -40
    db[7] = 0
    qs[::2] *= 2.0 ** -30
    got = ops.similarity(qs, db).double()
    ref = qs.double() @ db.double().t()
    scale = qs.double().abs() @ db.double().abs().t()
    rel = ((got - ref).abs() / scale.clamp_min(1e-300))
    rel[:, 7] = 0
    assert torch.isfinite(got).all() and float(got[:, 7].abs().max()) == 0.0
    assert float(rel.max()) < 3e-7, float(rel.max())


def test_split_similarity_ranks_like_the_exact_chain(tmp_path, monkeypatch):
    """The AP protocol over split-kernel scores equals the one over exact-chain scores when no two relevant
    scores sit within the rounding noise of each other (planted positives, 40k distractors)."""
    from dirtorch_amd import ops, ranking
    r = np.random.RandomState(8)
    N, Q, D = 40000, 12, 512
    db, gnd = make_db(tmp_path, N, Q, r, npos=30, njunk=8)
    g = torch.Generator(device='cuda').manual_seed(9)
    base = torch.randn(N, D, generator=g, device='cuda')
    qs = torch.randn(Q, D, generator=g, device='cuda')
    for q in range(Q):
        idx = torch.tensor(gnd[q]['easy'] + gnd[q]['hard'], device='cuda')
        base[idx] += qs[q] * torch.rand(len(idx), 1, generator=g, device='cuda') * 1.5
    base = torch.nn.functional.normalize(base, dim=1)
    qs = torch.nn.functional.normalize(qs, dim=1)
    split = ranking.eval_aps_device(db, ops.similarity(qs, base))
    monkeypatch.setenv('DIRTORCH_AMD_SIM_EXACT', '1')
    exact = ranking.eval_aps_device(db, ops.similarity(qs, base))
    for a, b in zip(split, exact):
        for m in ('easy', 'medium', 'hard'):
            assert a[m] == pytest.approx(b[m], abs=1e-4), m


@pytest.mark.parametrize('W', [2, 8])
def test_sharded_scoring_with_unequal_shards(W):
    """BASELINE configs[3] on 8 GPUs: 1 006 322 % 8 = 2, so the all-gathered database arrives as PADDED blocks with the
    padding rows between the shards, and bench.py / a multi-GPU eval score it block by block
    (dirtorch_amd.distributed.score_gathered).  Simulated on one GPU by slicing: N = 300 007 rows (N % W != 0, every
    shard still long enough for the split-bf16 similarity kernel), W = 2 and 8.
      * the two exchange layouts - descriptor blocks vs per-rank score blocks - give the same scores BIT FOR BIT (they
        run the same similarity call on the same rows);
      * against the un-sharded call the scores agree to fp32 summation order (sim_split rotates each row tile's
        starting K slab by its tile index, so a row's sum order depends on where its tile sits in the call: same sum,
        another association) and every AP to 1e-9."""
    from dirtorch_amd import distributed as dd
    from dirtorch_amd import ops, ranking
    N, Q, D = 300007, 24, 2048
    g = torch.Generator(device='cuda').manual_seed(7)
    db = torch.empty(N, D, device='cuda')
    for i in range(0, N, 65536):
        n = min(65536, N - i)
        db[i:i + n] = torch.nn.functional.normalize(torch.randn(n, D, generator=g, device='cuda'), dim=1)
    qs = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device='cuda'), dim=1)
    sizes, rows = dd.shard_sizes(N, W), dd.padded_rows(N, W)
    assert rows * W != N and min(h - l for l, h in sizes) >= 32768
    gathered = torch.zeros(W * rows, D, device='cuda')            # what all_gather_into_tensor of the padded blocks leaves
    for r, (l, h) in enumerate(sizes):
        gathered[r * rows:r * rows + (h - l)] = db[l:h]
    s_desc = dd.score_gathered(qs, gathered, N, W, ops.similarity)
    blocks = torch.stack([ops.similarity(qs, gathered[r * rows:(r + 1) * rows]) for r in range(W)])   # each rank's [Q, rows]
    s_score = dd.merge_score_blocks(blocks, N, W)
    assert s_desc.shape == (Q, N) and torch.equal(s_desc, s_score)
    s_one = ops.similarity(qs, db)
    assert float((s_desc - s_one).abs().max()) < 5e-7
    ref = (qs[:4].double() @ db.double().T)
    assert float((s_desc[:4].double() - ref).abs().max()) < 5e-7

    class _DB(object):
        relevants = None
    r = np.random.RandomState(3)
    d = _DB()
    d.nimg, d.nquery, d.easy, d.hard, d.junk = N, Q, [], [], []
    for q in range(Q):
        idx = r.choice(N, 60, replace=False)
        d.easy.append(sorted(idx[:20].tolist()))
        d.hard.append(sorted(idx[20:40].tolist()))
        d.junk.append(sorted(idx[40:].tolist()))
    t = ranking.build_probe_tables(d)
    a_sh, a_one = ranking.eval_aps_device(d, s_desc, t), ranking.eval_aps_device(d, s_one, t)
    for x, y in zip(a_sh, a_one):
        for k in x:
            assert abs(x[k] - y[k]) < 1e-9, (k, x[k], y[k])
