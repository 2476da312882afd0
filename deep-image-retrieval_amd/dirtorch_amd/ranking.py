"""Device-side ranking + AP for revisitop-style datasets.

Same numbers as `[db.eval_query_AP(q, s) for q, s in enumerate(scores)]`
(dirtorch/test_dir.py:153, dirtorch/datasets/generic.py:196-224) without downloading the Q x N score
matrix or sorting it: the kernel counts, for every listed image of every query, how many database
items rank before it; junk corrections and the AP sum touch only those few hundred listed images
and run on the host.  Ties rank by descending index (np.argsort(...)[::-1] with a stable order).
"""
import numpy as np
import torch

from . import ops
from .datasets import compute_average_precision


def similarity_device(qdescs, bdescs):
    """Scores Q.DB^T as a CUDA tensor [Q, N] (common.matmul without the download)."""
    from .utils.common import _dev
    return ops.gemm_nt(_dev(bdescs), _dev(qdescs))


def _before(sj, j, sp, p):
    return (sj > sp) | ((sj == sp) & (j > p))


def eval_aps_device(db, scores):
    """scores: CUDA tensor [Q, N].  Returns the list the reference builds: one float per query
    (classic protocol) or one {'easy','medium','hard'} dict per query."""
    Q, N = scores.shape
    assert Q == db.nquery and N == db.nimg, "scores should have shape (%d, %d)" % (db.nquery, db.nimg)
    classic = bool(db.relevants)
    lists = []
    for q in range(Q):
        if classic:
            groups = {'ok': list(db.relevants[q]), 'junk': list(db.junk[q])}
        else:
            groups = {'easy': list(db.easy[q]), 'hard': list(db.hard[q]), 'junk': list(db.junk[q])}
        lists.append(groups)
    P = max(1, max(sum(len(v) for v in g.values()) for g in lists))
    probe = -np.ones((Q, P), dtype=np.int32)
    for q, g in enumerate(lists):
        flat = [i for v in g.values() for i in v]
        probe[q, :len(flat)] = flat
    out_counts = np.zeros((Q, P), dtype=np.int64)
    out_scores = np.zeros((Q, P), dtype=np.float32)
    for p0 in range(0, P, 1024):          # the kernel takes up to 1024 probes per query per launch
        chunk = np.ascontiguousarray(probe[:, p0:p0 + 1024])
        c, s = ops.rank_counts(scores.contiguous(), torch.from_numpy(chunk).cuda())
        out_counts[:, p0:p0 + 1024] = c.cpu().numpy()
        out_scores[:, p0:p0 + 1024] = s.cpu().numpy()

    def ap(q, positives, junk):
        """AP with `positives` relevant and `junk` removed; -1 when there is no positive."""
        g = lists[q]
        flat = [i for v in g.values() for i in v]
        pos_of = {}
        for k, i in enumerate(flat):
            pos_of.setdefault(i, k)       # an index listed twice: any copy carries the same numbers
        if not positives:
            return -1
        # an image listed as positive AND junk is junk (the reference writes junk last, generic.py:192)
        junkset = set(junk)
        pos = [i for i in dict.fromkeys(positives) if i not in junkset]
        if not pos:
            return -1
        jk = np.array(sorted(junkset), dtype=np.int64)
        js = out_scores[q, [pos_of[i] for i in jk]] if len(jk) else np.zeros(0, np.float32)
        ranks = []
        for i in pos:
            k = pos_of[i]
            n_before = int(out_counts[q, k])
            if len(jk):
                n_before -= int(np.sum(_before(js, jk, out_scores[q, k], i)))
            ranks.append(n_before)
        return compute_average_precision(np.sort(np.array(ranks)))

    res = []
    for q, g in enumerate(lists):
        if classic:
            a = ap(q, g['ok'], g['junk'])
            res.append(0.0 if a == -1 else a)      # classic protocol has no -1 (generic.py:199-208)
        else:
            res.append({'easy': ap(q, g['easy'], g['junk'] + g['hard']),
                        'medium': ap(q, g['easy'] + g['hard'], g['junk']),
                        'hard': ap(q, g['hard'], g['junk'] + g['easy'])})
    return res
