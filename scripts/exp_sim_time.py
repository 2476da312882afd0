import os, sys, torch
sys.path.insert(0, 'deep-image-retrieval_amd')
from dirtorch_amd import ops
N, Q, D = 1006322, 70, 2048
g = torch.Generator(device='cuda').manual_seed(1)
db = torch.empty(N, D, device='cuda')
for i in range(0, N, 65536):
    n = min(65536, N - i)
    db[i:i+n] = torch.nn.functional.normalize(torch.randn(n, D, generator=g, device='cuda'), dim=1)
q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device='cuda'), dim=1)
for _ in range(2): ops.similarity(q, db)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    e0.record()
    for _ in range(5): ops.similarity(q, db)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
print(os.environ.get('DIRTORCH_AMD_LIB', 'default'), 'similarity 70 x 1006322 x 2048: %.3f ms' % best)
