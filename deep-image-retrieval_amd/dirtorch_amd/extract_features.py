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
    """Extract (pool, whiten) descriptors of a dataset and save them as .npy."""
    print("\n>> Extracting features...")
    try:
        query_db = db.get_query_db()
    except NotImplementedError:
        query_db = None

    bdescs, qdescs = [], []
    trfs_list = [trfs] if isinstance(trfs, str) else trfs
    for trfs in trfs_list:
        kw = dict(iscuda=net.iscuda, threads=threads, batch_size=batch_size,
                  same_size='Pad' in trfs or 'Crop' in trfs)
        bdescs.append(ddist.extract_sharded(test.extract_image_features, db, trfs, net, desc="DB", **kw))
        if query_db is not None:
            qdescs.append(bdescs[-1] if db is query_db
                          else test.extract_image_features(query_db, trfs, net, desc="query", **kw))

    bdescs = tonumpy(common.l2_normalize(pool(bdescs, pooling, gemp)))
    if query_db is not None:
        qdescs = tonumpy(common.l2_normalize(pool(qdescs, pooling, gemp)))

    if whiten is not None:
        bdescs = common.whiten_features(bdescs, net.pca, **whiten)
        if query_db is not None:
            qdescs = common.whiten_features(qdescs, net.pca, **whiten)

    if ddist.rank() == 0:
        mkdir(output, isfile=True)
        if query_db is db or query_db is None:
            np.save(output, bdescs)
        else:
            o = osp.splitext(output)
            np.save(o[0] + '.qdescs' + o[1], qdescs)
            np.save(o[0] + '.dbdescs' + o[1], bdescs)
        print('Features extracted.')
    return bdescs


load_model = test.load_model


def main(argv=None):
    parser = test.build_parser('Extract features')
    parser.add_argument('--output', type=str, default="", help='path to output features')
    parser.add_argument('--gpu', type=int, nargs='+', help='GPU ids')
    parser.add_argument('--whiten', type=str, default=None, help='applies whitening')
    parser.add_argument('--whitenp', type=float, default=0.5, help='whitening power, default is 0.5 (i.e., the sqrt)')
    args = parser.parse_args(argv)
    args.iscuda = test.setup_devices(args.gpu)

    dataset = datasets.create(args.dataset)
    print("Dataset:", dataset)

    net = load_model(args.checkpoint, args.iscuda)
    if args.whiten:
        net.pca = net.pca[args.whiten]
        args.whiten = {'whitenp': args.whitenp, 'whitenv': args.whitenv, 'whitenm': args.whitenm}
    else:
        net.pca = None
        args.whiten = None

    return extract_features(dataset, net, args.trfs, pooling=args.pooling, gemp=args.gemp, detailed=args.detailed,
                            threads=args.threads, dbg=args.dbg, whiten=args.whiten, output=args.output)


if __name__ == '__main__':
    main(sys.argv[1:])
