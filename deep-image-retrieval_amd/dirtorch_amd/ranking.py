"""Device-side ranking + AP for revisitop-style datasets.

Same numbers as `[db.eval_query_AP(q, s) for q, s in enumerate(scores)]`
(dirtorch/test_dir.py:153, dirtorch/datasets/generic.py:196-224) without downloading the Q x N score
matrix or sorting it: one kernel counts, for every listed image of every query, how many database
items rank before it; a second one applies the junk corrections among those few hundred listed images
and sums the AP in fp64 in the reference's order.  Ties rank by descending index (np.argsort(...)[::-1] with a stable order).
"""
import numpy as np
import torch

from . import ops


UNIT_RANGE_BOUND = 60.0     # csrc/sim_split.hip PAIR form: operands in (-64, 64)
UNIT_RANGE_MIN_ROWS = 32768  # ... which only exists on the large-database path


def is_unit_range(*tensors):
    """True when every value lies inside the fp16-pair similarity kernel's range (one abs-max pass per tensor, one
    host sync: call it once per database, not per query batch).  L2-normalised descriptors always qualify."""
    for t in tensors:
        if t.numel():
            lo, hi = torch.aminmax(t)      # (no |t| temporary: the database is 8 GB at config D's sizes)
            if not max(-float(lo), float(hi)) < UNIT_RANGE_BOUND:      # (NaN compares false: not in range)
                return False
    return True


_RANGE_ATTR = '_dirtorch_unit_range'     # (tensor._version, data_ptr, verdict), kept ON the caller's tensor object


def database_is_unit_range(b):
    """is_unit_range(b), remembered ON the database tensor the caller holds: the full pass over an 8 GB database (1.5 ms + a
    host sync) is paid once per tensor OBJECT, not on every query batch.  The verdict is an attribute of that object, so it
    dies with it - a new tensor that the caching allocator places at the same address starts without one (round-5 advice:
    a (data_ptr, shape, version) key could be served to a different upload) - and it carries the version counter and the
    address it was taken at, so an in-place torch update or a .set_() re-checks.  What torch cannot see - a refill through
    raw pointers, e.g. by this library's own kernels - must drop it: forget_unit_range(b), or pass unit_range= explicitly."""
    tag = getattr(b, _RANGE_ATTR, None)
    if tag is not None and tag[0] == b._version and tag[1] == b.data_ptr():
        return tag[2]
    verdict = is_unit_range(b)
    try:
        setattr(b, _RANGE_ATTR, (b._version, b.data_ptr(), verdict))
    except AttributeError:      # (an object that takes no attributes: no caching)
        pass
    return verdict


def forget_unit_range(b):
    """Drop the remembered range verdict of a database tensor whose contents were rewritten behind torch's back."""
    if hasattr(b, _RANGE_ATTR):
        delattr(b, _RANGE_ATTR)


def similarity_device(qdescs, bdescs, unit_range=None):
    """Scores Q.DB^T as a CUDA tensor [Q, N] (common.matmul without the download).  unit_range: True = the caller knows
    both sets are bounded by 60 in magnitude (L2-normalised descriptors: dirtorch/test_dir.py:150) and wants the fp16-pair
    kernel on large databases (ops.similarity); None (default) = look, when the database is large enough for it to matter -
    the DATABASE's verdict is remembered on the caller's own CUDA tensor (database_is_unit_range), only the small query block is checked per call;
    False = never.  Evaluation loops that know their descriptors are L2-normalised pass True."""
    from .utils.common import _dev
    q, b = _dev(qdescs), _dev(bdescs)
    if unit_range is None:
        # the verdict is only REMEMBERED for a database the caller owns as a float32 CUDA tensor (_dev hands that very object
        # back); an ndarray / a CPU or non-fp32 tensor is uploaded into a temporary, which is checked in full and forgotten
        owned = b is bdescs
        unit_range = (b.shape[0] >= UNIT_RANGE_MIN_ROWS and (database_is_unit_range(b) if owned else is_unit_range(b))
                      and is_unit_range(q))
    return ops.similarity(q, b, unit_range=bool(unit_range))


def _mode_lists(groups, classic):
    """[(positives, junk)] per mode, as the reference builds them (generic.py:150-170, 196-224)."""
    if classic:
        return [(groups['ok'], groups['junk'])]
    return [(groups['easy'], groups['junk'] + groups['hard']),
            (groups['easy'] + groups['hard'], groups['junk']),
            (groups['hard'], groups['junk'] + groups['easy'])]


def build_probe_tables(db):
    """Host-side index tables for eval_aps_device, built once per dataset: the union list of listed
    images per query (probe_idx [Q,P], -1 padded) and, per (query, mode), the positions of the mode's
    positives and junk inside that list (CSR).  Duplicates are dropped; an image listed as positive AND
    junk is junk (the reference writes junk last, generic.py:192)."""
    Q = db.nquery
    classic = bool(db.relevants)
    modes = 1 if classic else 3
    rows, pos_off, pos_list, junk_off, junk_list = [], [0], [], [0], []
    for q in range(Q):
        if classic:
            groups = {'ok': list(db.relevants[q]), 'junk': list(db.junk[q])}
        else:
            groups = {'easy': list(db.easy[q]), 'hard': list(db.hard[q]), 'junk': list(db.junk[q])}
        flat = list(dict.fromkeys(i for v in groups.values() for i in v))
        where = {i: k for k, i in enumerate(flat)}
        rows.append(flat)
        for positives, junk in _mode_lists(groups, classic):
            junkset = set(junk)
            pos_list += [where[i] for i in dict.fromkeys(positives) if i not in junkset]
            junk_list += [where[i] for i in sorted(junkset)]
            pos_off.append(len(pos_list))
            junk_off.append(len(junk_list))
    P = max(1, max(len(r) for r in rows))
    probe = -np.ones((Q, P), dtype=np.int32)
    for q, r in enumerate(rows):
        probe[q, :len(r)] = r
    as_dev = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int32)).cuda()   # noqa: E731
    return dict(probe=torch.from_numpy(probe).cuda(), pos_off=as_dev(pos_off), pos_list=as_dev(pos_list or [0]),
                junk_off=as_dev(junk_off), junk_list=as_dev(junk_list or [0]), modes=modes, classic=classic)


def eval_aps_device(db, scores, tables=None):
    """scores: CUDA tensor [Q, N].  Returns the list the reference builds: one float per query
    (classic protocol) or one {'easy','medium','hard'} dict per query.  Two kernels: the dense rank
    counts of every listed image (dir_rank_counts, one pass over the score rows) and the junk-corrected
    APs (dir_revisitop_ap); only Q x modes doubles come back to the host."""
    Q, N = scores.shape
    assert Q == db.nquery and N == db.nimg, "scores should have shape (%d, %d)" % (db.nquery, db.nimg)
    t = tables if tables is not None else build_probe_tables(db)
    counts, pscores = ops.rank_counts(scores.contiguous(), t['probe'])
    ap = ops.revisitop_ap(t['probe'], counts, pscores, t['pos_off'], t['pos_list'], t['junk_off'],
                          t['junk_list'], t['modes']).cpu().numpy()
    if t['classic']:
        return [0.0 if a == -1 else float(a) for a in ap[:, 0]]     # classic protocol has no -1 (generic.py:199-208)
    return [{'easy': float(a[0]) if a[0] != -1 else -1, 'medium': float(a[1]) if a[1] != -1 else -1,
             'hard': float(a[2]) if a[2] != -1 else -1} for a in ap]
