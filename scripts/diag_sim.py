import os, sys, pickle, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-image-retrieval_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
from test_ranking_gpu import make_db
from dirtorch_amd import ranking, ops
r = np.random.RandomState(2)
Nb, Nd, Q, D = 6322, 1000000, 70, 2048
N = Nb + Nd
tmp = tempfile.mkdtemp()
db, gnd = make_db(tmp, N, Q, r, npos=40, njunk=10)
g = torch.Generator(device='cuda').manual_seed(3)
base = torch.empty(N, D, device='cuda')
for i in range(0, N, 131072):
    base[i:i + 131072] = torch.randn(min(131072, N - i), D, generator=g, device='cuda')
qs = torch.randn(Q, D, generator=g, device='cuda')
for q in range(Q):
    idx = torch.tensor(gnd[q]['easy'] + gnd[q]['hard'], device='cuda')
    base[idx] += qs[q] * torch.rand(len(idx), 1, generator=g, device='cuda') * 1.5
for i in range(0, N, 131072):
    base[i:i + 131072] = torch.nn.functional.normalize(base[i:i + 131072], dim=1)
qs = torch.nn.functional.normalize(qs, dim=1)
for mode in ('split', 'exact'):
    if mode == 'exact': os.environ['DIRTORCH_AMD_SIM_EXACT'] = '1'
    scores = ranking.similarity_device(qs, base)
    dev = ranking.eval_aps_device(db, scores)
    sc = scores.cpu().numpy()
    for q in (0, 1, 2, 35, 69):
        host = db.eval_query_AP(q, sc[q])
        pos = gnd[q]['easy'] + gnd[q]['hard'] + gnd[q]['junk']
        ties = [(p, int((sc[q] == sc[q][p]).sum())) for p in pos if (sc[q] == sc[q][p]).sum() > 1]
        print(mode, q, {m: (dev[q][m], host[m]) for m in ('easy', 'medium', 'hard') if abs(dev[q][m] - host[m]) > 1e-12}, 'ties among listed:', ties)
