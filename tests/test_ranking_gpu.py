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
    scores = r.standard_normal((Q, N)).astype(np.float32)       # continuous: no ties
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
    """BASELINE config D scale on one GPU: 70 queries x (6322 + 1e6 distractors), D = 128 here to
    keep the CPU cross-check cheap.  Properties: device AP == host AP on the same device scores for
    sampled queries; adding distractors can only lower AP."""
    from dirtorch_amd import ranking
    r = np.random.RandomState(2)
    Nb, Nd, Q, D = 6322, 1000000, 70, 128
    db, gnd = make_db(tmp_path, Nb + Nd, Q, r, npos=40, njunk=10)
    g = torch.Generator(device='cuda').manual_seed(3)
    base = torch.randn(Nb + Nd, D, generator=g, device='cuda')
    qs = torch.randn(Q, D, generator=g, device='cuda')
    for q in range(Q):                                   # plant the positives near their query
        idx = torch.tensor(gnd[q]['easy'] + gnd[q]['hard'], device='cuda')
        base[idx] += qs[q] * torch.rand(len(idx), 1, generator=g, device='cuda') * 1.5
    base = torch.nn.functional.normalize(base, dim=1)
    qs = torch.nn.functional.normalize(qs, dim=1)
    scores = ranking.similarity_device(qs, base)
    assert scores.shape == (Q, Nb + Nd)
    dev = ranking.eval_aps_device(db, scores)
    sc = scores.cpu().numpy()
    for q in (0, 1, 2, 35, 69):
        host = db.eval_query_AP(q, sc[q])
        for m in ('easy', 'medium', 'hard'):
            assert dev[q][m] == pytest.approx(host[m], abs=1e-12), (q, m)
    med = np.mean([d['medium'] for d in dev if d['medium'] >= 0])
    assert 0.05 < med <= 1.0
