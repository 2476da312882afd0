"""Feature-dump entry point - same function and flags as dirtorch/extract_features.py.

    python -m dirtorch_amd.extract_features --dataset 'ImageList("list.txt")' --checkpoint X.pt \
        --output feats.npy --gpu 0 [--whiten Landmarks_clean --whitenp 0.5]

Writes <output> (or <output>.qdescs / .dbdescs when the dataset has a separate query set),
extract_features.py:26-68.
"""
import os.path as osp
import sys

import numpy as np

from . import datasets
from . import distributed as ddist
from . import test_dir as test
from .utils import common
from .utils.common import pool, tonumpy
from .utils.convenient import mkdir


def extract_features(db, net, trfs, pooling='mean', gemp=3, detailed=False, whiten=None,
                     threads=8, batch_size=16, output=None, dbg=()):
    """Descriptors of `db` (pooled over the transform chains, L2-normalised, optionally whitened)
    saved as .npy: one file, or <output>.qdescs / .dbdescs when the dataset has its own query set."""
    print("\n>> Extracting features...")
    try:
        query_db = db.get_query_db()
    except NotImplementedError:
        query_db = None
    separate_queries = query_db is not None and query_db is not db

    kw = dict(threads=threads, batch_size=batch_size)
    per_scale = {'db': test.extract_per_scale(db, trfs, net, desc="DB", sharded=True, **kw),
                 'q': test.extract_per_scale(query_db, trfs, net, desc="query", **kw) if separate_queries else []}

    def finish(descs):
        descs = tonumpy(common.l2_normalize(pool(descs, pooling, gemp)))
        return common.whiten_features(descs, net.pca, **whiten) if whiten is not None else descs

    bdescs = finish(per_scale['db'])
    qdescs = finish(per_scale['q']) if separate_queries else None

    if ddist.rank() == 0:
        mkdir(output, isfile=True)
        if separate_queries:
            stem, ext = osp.splitext(output)
            np.save(stem + '.qdescs' + ext, qdescs)
            np.save(stem + '.dbdescs' + ext, bdescs)
        else:
            np.save(output, bdescs)
        print('Features extracted.')
    return bdescs


load_model = test.load_model


def main(argv=None):
    args = test.build_parser('Extract features', extra=[
        (('--output',), dict(type=str, default='', help='path to output features')),
        (('--gpu',), dict(type=int, nargs='+', help='GPU ids')),
        (('--whiten',), dict(type=str, default=None, help='applies whitening')),
        (('--whitenp',), dict(type=float, default=0.5, help='whitening power, default is 0.5 (i.e., the sqrt)')),
    ]).parse_args(argv)
    iscuda = test.setup_devices(args.gpu)
    dataset = datasets.create(args.dataset)
    print("Dataset:", dataset)
    net = load_model(args.checkpoint, iscuda)
    whiten = test.select_whitening(net, args)
    return extract_features(dataset, net, args.trfs, pooling=args.pooling, gemp=args.gemp,
                            detailed=args.detailed, threads=args.threads, dbg=args.dbg, whiten=whiten,
                            output=args.output)


if __name__ == '__main__':
    main(sys.argv[1:])
