#!/usr/bin/env python
"""Timing of the post-extraction path at BASELINE config C / D sizes on one MI355X:
PCA whitening, Q x N similarity (fp32 MFMA), device-side rank counts + host AP.
    python scripts/bench_rank.py            # prints one JSON line
DB descriptors are unit-norm randn rows generated on the device."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-image-retrieval_amd'))
import numpy as np
import torch
from dirtorch_amd import _lib, ops, ranking
from dirtorch_amd.datasets import ImageListRelevants


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


class FakeDB(object):
    relevants = None

    def __init__(self, N, Q, r):
        self.nimg, self.nquery = N, Q
        self.easy, self.hard, self.junk = [], [], []
        for q in range(Q):
            idx = r.choice(N, 200, replace=False)
            self.easy.append(sorted(idx[:80].tolist()))
            self.hard.append(sorted(idx[80:160].tolist()))
            self.junk.append(sorted(idx[160:].tolist()))


def main():
    res = {}
    g = torch.Generator(device='cuda').manual_seed(2)
    D, Q = 2048, 70
    for tag, N in (('roxford5k', 4993), ('rparis6k+1M', 1006322)):
        db = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device='cuda'), dim=1)
        qs = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device='cuda'), dim=1)
        ms, scores = timed(lambda: ops.similarity(qs, db))
        if N >= 32768:   # the exact fp32 MFMA chain on the same data, for the record
            os.environ["DIRTORCH_AMD_SIM_EXACT"] = "1"
            _lib.reload_env()       # (the library reads its switches once)
            ms_exact, exact = timed(lambda: ops.similarity(qs, db))
            del os.environ["DIRTORCH_AMD_SIM_EXACT"]
            _lib.reload_env()
        res[tag] = {'N': N, 'similarity_ms': round(ms, 3),
                    'similarity_GBps': round(N * D * 4 / ms / 1e6, 1),
                    'similarity_TFLOPs': round(2.0 * Q * N * D / ms / 1e9, 2)}
        if N >= 32768:
            res[tag]['similarity_exact_fp32_ms'] = round(ms_exact, 3)
            res[tag]['split_vs_exact_max_abs'] = float((scores - exact).abs().max())
            del exact
        fdb = FakeDB(N, Q, np.random.RandomState(1))
        t0 = time.perf_counter()
        aps = ranking.eval_aps_device(fdb, scores)
        torch.cuda.synchronize()
        res[tag]['rank_ap_ms'] = round((time.perf_counter() - t0) * 1e3, 2)
        if N < 100000:   # CPU reference timing of the same step (np.dot + 3 argsorts per query)
            sc = scores.cpu().numpy()
            t0 = time.perf_counter()
            q_np, db_np = qs.cpu().numpy(), db.cpu().numpy()
            np.dot(q_np, db_np.T)
            for q in range(Q):
                np.argsort(sc[q])[::-1]
            res[tag]['cpu_dot_plus_argsort_ms'] = round((time.perf_counter() - t0) * 1e3, 2)
        del db, scores
    # PCA whitening of 100k descriptors to 2048 components
    N = 100000
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device='cuda'), dim=1)
    C = torch.randn(D, D, generator=g, device='cuda') / 45.0
    mean = X.mean(0).contiguous()
    alpha = torch.rand(D, generator=g, device='cuda') + 0.5
    ms, _ = timed(lambda: ops.l2norm_rows_(ops.gemm_nt(C, X, qsub=mean, alpha=alpha), 0.0), reps=3)
    res['whiten_100k'] = {'ms': round(ms, 2), 'TFLOPs_fp32': round(2.0 * N * D * D / ms / 1e9, 1)}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
