"""Evaluation entry point - same functions and flags as dirtorch/test_dir.py.

    python -m dirtorch_amd.test_dir --dataset ROxford5K --checkpoint X.pt --whiten Landmarks_clean \
        --whitenp 0.25 --gpu 0

    expand_descriptors        test_dir.py:24-44    (alpha query expansion / DB augmentation)
    extract_image_features    test_dir.py:47-94    (the hot loop: loader -> net -> descriptors)
    eval_model                test_dir.py:97-180   (extract -> pool -> whiten -> scores -> AP)
    load_model                test_dir.py:183-191

All arithmetic runs on the engine (dirtorch_amd.nets / dirtorch_amd.utils.common); this file is
orchestration only.  Under torch.distributed (one process per GPU) the database is sharded
image-parallel and gathered once before ranking (dirtorch_amd.distributed).
"""
import json
import os
import sys

import numpy as np
import torch
import tqdm

from . import datasets
from . import distributed as ddist
from . import nets
from .utils import common
from .utils.common import matmul, pool, tonumpy
from .utils.convenient import mkdir
from .utils.pytorch_loader import get_loader


def expand_descriptors(descs, db=None, alpha=0, k=0):
    """alpha-weighted query expansion (db given) or database-side augmentation (db=None)."""
    assert k >= 0 and alpha >= 0, 'k and alpha must be non-negative'
    if k == 0:
        return descs
    descs = tonumpy(descs)
    n = descs.shape[0]
    db_descs = tonumpy(db if db is not None else descs)

    sim = matmul(descs, db_descs)            # fp32 MFMA similarity kernel
    if db is None:
        sim[np.diag_indices(n)] = 0

    idx = np.argpartition(sim, int(-k), axis=1)[:, int(-k):]
    descs_aug = np.zeros_like(descs)
    for i in range(n):
        new_q = np.vstack([db_descs[j, :] * sim[i, j] ** alpha for j in idx[i]])
        new_q = np.vstack([descs[i], new_q])
        new_q = np.mean(new_q, axis=0)
        descs_aug[i] = new_q / np.linalg.norm(new_q)
    return descs_aug


def extract_image_features(dataset, transforms, net, ret_imgs=False, same_size=False, flip=None,
                           desc="Extract feats...", iscuda=True, threads=8, batch_size=8):
    """Descriptors of every image of `dataset` -> Tensor [N, D] on the net's device.
    Variable-size images force batch_size 1, as in the reference (test_dir.py:52-55)."""
    if not same_size:
        batch_size = 1

    loader = get_loader(dataset, trf_chain=transforms, preprocess=net.preprocess, iscuda=iscuda,
                        output=['img'], batch_size=batch_size, threads=threads, shuffle=False)
    if hasattr(net, 'eval'):
        net.eval()

    tocpu = (lambda x: x.cpu()) if ret_imgs == 'cpu' else (lambda x: x)
    img_feats, trf_images = [], []
    with torch.no_grad():
        for inputs in tqdm.tqdm(loader, desc, total=1 + (len(dataset) - 1) // batch_size):
            imgs = inputs[0]
            wdim = 2 if imgs.dtype == torch.uint8 else 3        # NHWC uint8 | NCHW float
            for i in range(len(imgs)):
                if flip and flip.pop(0):
                    imgs[i] = imgs[i].flip(wdim - 1)
            imgs = common.variables(inputs[:1], net.iscuda)[0]
            d = net(imgs)
            if ret_imgs:
                trf_images.append(tocpu(imgs.detach()))
            del imgs, inputs
            if len(d.shape) == 1:
                d = d.unsqueeze(0)
            img_feats.append(d.detach())

    img_feats = torch.cat(img_feats, dim=0)
    if len(img_feats.shape) == 1:
        img_feats = img_feats.unsqueeze(0)
    if ret_imgs:
        if same_size:
            trf_images = torch.cat(trf_images, dim=0)
        return trf_images, img_feats
    return img_feats


def eval_model(db, net, trfs, pooling='mean', gemp=3, detailed=False, whiten=None,
               aqe=None, adba=None, threads=8, batch_size=16, save_feats=None,
               load_feats=None, dbg=()):
    """Evaluate a network on a retrieval dataset that carries its own AP protocol."""
    print("\n>> Evaluation...")
    query_db = db.get_query_db()

    bdescs, qdescs = [], []
    if not load_feats:
        trfs_list = [trfs] if isinstance(trfs, str) else trfs
        for trfs in trfs_list:
            kw = dict(iscuda=net.iscuda, threads=threads, batch_size=batch_size,
                      same_size='Pad' in trfs or 'Crop' in trfs)
            # image-parallel shards + one all-gather when torch.distributed is initialised
            bdescs.append(ddist.extract_sharded(extract_image_features, db, trfs, net, desc="DB", **kw))
            qdescs.append(bdescs[-1] if db is query_db
                          else extract_image_features(query_db, trfs, net, desc="query", **kw))
        # pool over transforms (scales), then L2
        bdescs = common.l2_normalize(pool(bdescs, pooling, gemp))
        qdescs = common.l2_normalize(pool(qdescs, pooling, gemp))
    else:
        bdescs = np.load(os.path.join(load_feats, 'feats.bdescs.npy'))
        qdescs = np.load(os.path.join(load_feats, 'feats.qdescs.npy')) if query_db is not db else bdescs

    if save_feats:
        mkdir(save_feats)
        np.save(os.path.join(save_feats, 'feats.bdescs.npy'), tonumpy(bdescs))
        if query_db is not db:
            np.save(os.path.join(save_feats, 'feats.qdescs.npy'), tonumpy(qdescs))

    if whiten is not None:
        bdescs = common.whiten_features(tonumpy(bdescs), net.pca, **whiten)
        qdescs = common.whiten_features(tonumpy(qdescs), net.pca, **whiten)

    # (the reference reads a module-global `args` here, test_dir.py:141,143; the parameters are meant)
    if adba is not None:
        bdescs = expand_descriptors(bdescs, **adba)
    if aqe is not None:
        qdescs = expand_descriptors(qdescs, db=bdescs, **aqe)

    # Large databases (>= 50k images, or DIRTORCH_AMD_DEVICE_RANK=1): keep the score matrix on the
    # GPU and rank there (dirtorch_amd.ranking) instead of downloading it and argsort-ing every row.
    flag = os.environ.get('DIRTORCH_AMD_DEVICE_RANK', 'auto')
    device_rank = hasattr(db, 'junk') and (flag == '1' or (flag == 'auto' and len(db) >= 50000))
    if device_rank:
        from . import ranking
        scores_dev = ranking.similarity_device(qdescs, bdescs)
        scores = []          # no per-row host scores: top-k below is skipped like for label-less sets
    else:
        scores = matmul(qdescs, bdescs)
    del bdescs, qdescs

    res = {}
    try:
        if device_rank:
            aps = ranking.eval_aps_device(db, scores_dev)
        else:
            aps = [db.eval_query_AP(q, s) for q, s in enumerate(tqdm.tqdm(scores, desc='AP'))]
        if not isinstance(aps[0], dict):
            aps = [float(e) for e in aps]
            if detailed:
                res['APs'] = aps
            res['mAP'] = float(np.mean([e for e in aps if e >= 0]))   # AP -1 = query without relevants
        else:
            for mode in aps[0].keys():
                apst = [float(e[mode]) for e in aps]
                if detailed:
                    res['APs' + '-' + mode] = apst
                res['mAP' + '-' + mode] = float(np.mean([e for e in apst if e >= 0]))
    except NotImplementedError:
        print(" AP not implemented!")

    try:
        if device_rank:
            raise NotImplementedError()   # the revisitop datasets carry no labels (dataset.py:97)
        tops = [db.eval_query_top(q, s) for q, s in enumerate(tqdm.tqdm(scores, desc='top1'))]
        if detailed:
            res['tops'] = tops
        for k in tops[0]:
            res['top%d' % k] = float(np.mean([top[k] for top in tops]))
    except NotImplementedError:
        pass
    return res


def load_model(path, iscuda):
    checkpoint = common.load_checkpoint(path, iscuda)
    net = nets.create_model(pretrained="", **checkpoint['model_options'])
    net = common.switch_model_to_cuda(net, iscuda, checkpoint)
    net.load_state_dict(checkpoint['state_dict'])
    net.preprocess = checkpoint.get('preprocess', net.preprocess)
    if 'pca' in checkpoint:
        net.pca = checkpoint.get('pca')
    return net


def setup_devices(gpus):
    """--gpu N (the reference's flag, common.torch_set_gpu) for a single process; under
    torch.distributed.run the launcher's LOCAL_RANK picks the GPU and the process group is joined."""
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        ddist.init_from_env()
        return True
    if gpus is None:
        gpus = [0]
    return common.torch_set_gpu(gpus)


def build_parser(description='Evaluate a model'):
    import argparse
    parser = argparse.ArgumentParser(description=description)
    parser.add_argument('--dataset', '-d', type=str, required=True, help='Command to load dataset')
    parser.add_argument('--checkpoint', type=str, required=True, help='path to weights')
    parser.add_argument('--trfs', type=str, required=False, default='', nargs='+', help='test transforms (can be several)')
    parser.add_argument('--pooling', type=str, default="gem", help='pooling scheme if several trf chains')
    parser.add_argument('--gemp', type=int, default=3, help='GeM pooling power')
    parser.add_argument('--out-json', type=str, default="", help='path to output json')
    parser.add_argument('--detailed', action='store_true', help='return detailed evaluation')
    parser.add_argument('--threads', type=int, default=8, help='number of thread workers')
    parser.add_argument('--dbg', default=(), nargs='*', help='debugging options')
    parser.add_argument('--whitenv', type=int, default=None, help='number of components, default is None (i.e. all components)')
    parser.add_argument('--whitenm', type=float, default=1.0, help='whitening multiplier, default is 1.0 (i.e. no multiplication)')
    return parser


def main(argv=None):
    parser = build_parser()
    parser.add_argument('--save-feats', type=str, default="", help='path to output features')
    parser.add_argument('--load-feats', type=str, default="", help='path to load features from')
    parser.add_argument('--gpu', type=int, default=0, nargs='+', help='GPU ids')
    parser.add_argument('--whiten', type=str, default='Landmarks_clean', help='applies whitening')
    parser.add_argument('--aqe', type=int, nargs='+', help='alpha-query expansion paramenters')
    parser.add_argument('--adba', type=int, nargs='+', help='alpha-database augmentation paramenters')
    parser.add_argument('--whitenp', type=float, default=0.25, help='whitening power, default is 0.5 (i.e., the sqrt)')
    args = parser.parse_args(argv)
    args.iscuda = setup_devices(args.gpu)
    if args.aqe is not None:
        args.aqe = {'k': args.aqe[0], 'alpha': args.aqe[1]}
    if args.adba is not None:
        args.adba = {'k': args.adba[0], 'alpha': args.adba[1]}

    dataset = datasets.create(args.dataset)
    print("Test dataset:", dataset)

    net = load_model(args.checkpoint, args.iscuda)
    if args.whiten:
        net.pca = net.pca[args.whiten]
        args.whiten = {'whitenp': args.whitenp, 'whitenv': args.whitenv, 'whitenm': args.whitenm}
    else:
        net.pca = None
        args.whiten = None

    res = eval_model(dataset, net, args.trfs, pooling=args.pooling, gemp=args.gemp, detailed=args.detailed,
                     threads=args.threads, dbg=args.dbg, whiten=args.whiten, aqe=args.aqe, adba=args.adba,
                     save_feats=args.save_feats, load_feats=args.load_feats)
    if ddist.rank() == 0:
        # (--detailed adds per-query lists; the reference's '%g' print dies on them)
        print(' * ' + '\n * '.join(['%s = %g' % p for p in res.items() if np.isscalar(p[1])]))
        if args.out_json:
            try:
                data = json.load(open(args.out_json))
            except IOError:
                data = {}
            data[args.dataset] = res
            mkdir(args.out_json, isfile=True)
            open(args.out_json, 'w').write(json.dumps(data, indent=1))
            print("saved to " + args.out_json)
    return res


if __name__ == '__main__':
    main(sys.argv[1:])
