"""Retrieval / classification metrics with the names of dirtorch/utils/evaluation.py.

    compute_average_precision   :46-82   revisited Oxford/Paris AP from the positives' ranks
    compute_AP                  :41-43   sklearn average_precision_score (labelled datasets)
    compute_average_precision_quantized :85-98   max-precision-above-recall AP on a recall grid
    accuracy_topk               :8-38    precision@k of a score matrix against integer labels

Host-side bookkeeping over a few hundred numbers per query; the Q x N arithmetic that feeds it
(similarity, rank counts) runs in the engine's kernels.
"""
import numpy as np
import torch


def compute_average_precision(positive_ranks):
    """Trapezoidal AP: `positive_ranks` are the sorted zero-based ranks of the positives among the
    non-junk images; an empty list scores 0."""
    n = len(positive_ranks)
    if not n:
        return 0.0
    ap = 0.0
    for i, rank in enumerate(positive_ranks):
        left = 1.0 if not rank else i / rank
        ap += (left + (i + 1) / (rank + 1)) / (2.0 * n)
    return ap


def compute_AP(label, score):
    from sklearn.metrics import average_precision_score
    return average_precision_score(label, score)


def compute_average_precision_quantized(labels, idx, step=0.01):
    """Mean over recall levels r = 0, step, 2*step, ... < 1 of the best precision reached at a recall
    above r, for the ranking `idx` of the binary `labels`; 0 when nothing is relevant."""
    labels = np.asarray(labels)
    nrel = labels.sum()
    if nrel == 0:
        return 0
    hits = np.cumsum(labels[idx])
    recall = hits / float(nrel)
    prec = hits.astype(np.float32) / np.arange(1, len(idx) + 1)
    levels = np.arange(0, 1, step)
    best = [prec[recall > r].max() if (recall > r).any() else 0 for r in levels]
    return np.mean(np.array(best))


def accuracy_topk(output, target, topk=(1,)):
    """Fraction of rows of `output` ([B, L] scores, NumPy or torch) whose true label `target[b]` is
    among the k best-scored labels, for every k of `topk`."""
    if isinstance(output, np.ndarray):
        order = np.argsort(-output, axis=1)
        hit = order == np.expand_dims(target, axis=1)
        return [hit[:, :k].sum() / target.size for k in topk]
    if isinstance(output, torch.Tensor):
        order = output.topk(max(topk), dim=1, largest=True, sorted=True)[1]
        hit = order.eq(target.unsqueeze(1))
        return [hit[:, :k].float().reshape(-1).sum(0) * (1.0 / target.numel()) for k in topk]
    raise TypeError('accuracy_topk: ndarray or Tensor expected, got %s' % type(output).__name__)
