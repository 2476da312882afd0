"""Retrieval evaluation on the MI355X engine - the functions and flags of dirtorch/test_dir.py.

    python -m dirtorch_amd.test_dir --dataset ROxford5K --checkpoint X.pt --whiten Landmarks_clean \
        --whitenp 0.25 --gpu 0

    expand_descriptors        test_dir.py:24-44    alpha query expansion / database augmentation
    extract_image_features    test_dir.py:47-94    the hot loop: loader -> net -> [N, D] descriptors
    eval_model                test_dir.py:97-180   extract -> pool -> whiten -> (QE) -> scores -> AP
    load_model                test_dir.py:183-191  checkpoint -> engine-backed network

This file only orchestrates: descriptors, pooling, whitening, similarity and (for large databases)
ranking are kernels of libdir_engine.so reached through dirtorch_amd.nets / .utils.common / .ranking.
Under torch.distributed (one process per GPU) the database is sharded image-parallel and gathered
once before ranking (dirtorch_amd.distributed).
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
from .utils import transforms as trf_mod
from .utils.pytorch_loader import get_loader


def expand_descriptors(descs, db=None, alpha=0, k=0):
    """alpha query expansion / database augmentation (test_dir.py:24-44): each descriptor becomes the
    L2-normalised mean of itself and its k nearest neighbours in `db` (or among the OTHER rows of
    `descs`), each neighbour weighted by similarity**alpha.  Similarity, top-k selection, weighted
    mean and norm all run on the GPU (dir_expand_descriptors); ndarray in, ndarray out like the
    reference.  Ties at the k-th similarity are broken by the larger index (np.argpartition leaves
    them unspecified)."""
    assert k >= 0 and alpha >= 0, 'k and alpha must be non-negative'
    if k == 0:
        return descs
    descs = tonumpy(descs)
    pool_ = None if db is None else tonumpy(db)
    m = len(descs) if pool_ is None else len(pool_)
    if int(k) > m:
        raise ValueError('kth(=%d) out of bounds (%d)' % (m - int(k), m))    # np.argpartition's error
    from . import ops
    out = ops.expand_descriptors(common._dev(descs), None if pool_ is None else common._dev(pool_),
                                 alpha=float(alpha), k=int(k))
    return out.cpu().numpy().astype(descs.dtype, copy=False)


def _check_finite(feats, net):
    """fp16 activations overflow at 65504 (the reference computes in fp32 and cannot).  Two checks, one
    sync per extraction pass: the engine's overflow word - every kernel that stores an fp16 inf / NaN
    sets it, so an overflow deep inside the trunk that a later ReLU flushes to zero is still caught -
    and the descriptors themselves (inf / NaN weights in the checkpoint show up here in any dtype)."""
    overflow = getattr(net, 'overflowed', lambda: False)()
    if overflow or (feats.numel() and not bool(torch.isfinite(feats).all())):
        dtype = getattr(net, 'compute_dtype', '?')
        if overflow or dtype in ('fp16', 'fp16p'):
            raise FloatingPointError('%s with compute dtype %s: activations left the fp16 range; run with '
                                     'DIRTORCH_AMD_DTYPE=bf16 (fp32 range, 8-bit mantissa) or DIRTORCH_AMD_DTYPE=f32 '
                                     '(the reference\'s arithmetic)'
                                     % ('fp16 overflow inside the trunk' if overflow else 'non-finite descriptors', dtype))
        # bf16 / f32 share fp32's range: inf / NaN here come from the checkpoint or the input, not from the format
        raise FloatingPointError('non-finite descriptors with compute dtype %s: the checkpoint or the input holds '
                                 'inf / NaN values (this format has fp32\'s range)' % dtype)
    return feats


class StreamPool(object):
    """Round-robin HIP streams for forwards that cannot share a batch (the reference's real workload: batch 1 at
    native resolution, test_dir.py:52-55).  One 1024^2 image gives layers 3 / 4 16-128 workgroups for 256 CUs;
    forwards issued on a few streams overlap on the device (each stream has its own engine workspace,
    nets/rmac_resnet.py _workspace): 770 -> 1325 img/s at 1024^2 and 825 -> 1700 at 768x1024 with four streams, the
    best of 1-6.  DIRTORCH_AMD_STREAMS=n (default 4; 1 = everything on the current stream).
    Results are bit-identical to the single-stream ones: the same kernels run on the same data, only their
    interleaving changes (scripts/exp_stream_race.py: 0 of 7 680 forwards differ; that script is also how the one
    kernel whose emitted code depended on timing was found - csrc/dir_common.h ring_barrier)."""

    _shared = {}      # (device, n) -> the process-wide streams of that pool size

    def __init__(self, n=None):
        n = int(os.environ.get('DIRTORCH_AMD_STREAMS', '4')) if n is None else n
        self.streams = []
        if n > 1 and torch.cuda.is_available():
            # ONE set of streams per device and pool size for the whole process: the engine keeps a workspace per stream
            # (nets/rmac_resnet.py _workspace, 0.2 ... 1+ GB each), so a fresh set of streams per extraction call - torch
            # hands out 32 distinct ones before it wraps - would leave up to 33 workspaces allocated over an evaluation
            key = (torch.cuda.current_device(), n)
            if key not in StreamPool._shared:
                StreamPool._shared[key] = [torch.cuda.Stream() for _ in range(n)]
            self.streams = StreamPool._shared[key]
        self.i = 0

    def run(self, fn, *inputs):
        """fn() on the next stream of the pool; `inputs` are the tensors it reads (produced on the current stream)."""
        if not self.streams:
            return fn()
        cur = torch.cuda.current_stream()
        s = self.streams[self.i % len(self.streams)]
        self.i += 1
        s.wait_stream(cur)                      # the inputs are ready when this stream starts
        for t in inputs:
            t.record_stream(s)                  # ... and stay allocated until it is done with them
        with torch.cuda.stream(s):
            out = fn()
        out.record_stream(cur)                  # consumed on the caller's stream after join()
        return out

    def join(self):
        cur = torch.cuda.current_stream()
        for s in self.streams:
            cur.wait_stream(s)


def extract_image_features(dataset, transforms, net, ret_imgs=False, same_size=False, flip=None,
                           desc="Extract feats...", iscuda=True, threads=8, batch_size=8):
    """One descriptor per image of `dataset`, as a [N, D] tensor on the network's device.
    Images of different sizes cannot share a batch: unless `same_size`, batch_size is forced to 1
    (the reference's real workload, test_dir.py:52-55)."""
    bs = batch_size if same_size else 1
    loader = get_loader(dataset, trf_chain=transforms, preprocess=net.preprocess, iscuda=iscuda,
                        output=['img'], batch_size=bs, threads=threads, shuffle=False)
    net.eval()
    if (not same_size and not ret_imgs and not flip and batch_size > 1 and net.iscuda
            and os.environ.get('DIRTORCH_AMD_BUCKET_BATCH', '1') != '0'):
        return _check_finite(_extract_bucketed(loader, len(dataset), net, batch_size, desc), net)
    feats, kept = [], []
    nbatches = (len(dataset) + bs - 1) // bs
    # (per-launch profiling records events of ONE stream in issue order: overlapping forwards would interleave them)
    pool_ = StreamPool(None if (bs == 1 and net.iscuda and not getattr(net, '_profiling', False)) else 1)
    with torch.no_grad():
        for (imgs,) in tqdm.tqdm(loader, desc, total=nbatches):
            if flip:
                waxis = 1 if imgs.dtype == torch.uint8 else 2     # width axis of one HWC / CHW image
                for i in range(len(imgs)):
                    if flip and flip.pop(0):
                        imgs[i] = imgs[i].flip(waxis)
            imgs = common.variables([imgs], net.iscuda)[0]
            d = pool_.run(lambda: net(imgs), imgs)
            feats.append(d.reshape(1, -1) if d.dim() == 1 else d)   # B == 1 comes back as [D]
            if ret_imgs:
                kept.append(imgs.cpu() if ret_imgs == 'cpu' else imgs)
    pool_.join()
    feats = _check_finite(torch.cat(feats, dim=0), net)
    if not ret_imgs:
        return feats
    return (torch.cat(kept, dim=0) if same_size else kept), feats


def _extract_bucketed(loader, n, net, batch_size, desc):
    """Variable-size images, batched anyway: the loader still yields one image at a time (in order),
    images are parked on the GPU in per-size buckets, a bucket runs as one batch when it holds
    `batch_size` images (or at the end), and every descriptor lands at its image's index.  The
    reference runs these datasets at batch 1 (test_dir.py:52-55); on the MI355X a batch of 8 same-size
    1024^2 images runs 2.3x faster per image than batch 1.  DIRTORCH_AMD_BUCKET_BATCH=0 disables."""
    out = None
    buckets = {}                                   # (H, W, dtype) -> ([indices], [image tensors])
    pending, cap = 0, 8 * batch_size               # bound the parked images
    pool_ = StreamPool()                           # part-filled buckets of rare sizes overlap on the device
    results = []                                   # (indices, descriptors), scattered after the streams joined

    def run(key):
        nonlocal pending
        idx, imgs = buckets.pop(key)
        x = torch.cat(imgs, dim=0)
        d = pool_.run(lambda: net(x).reshape(len(idx), -1), x)
        results.append((idx, d))
        pending -= len(idx)

    def scatter():
        nonlocal out
        pool_.join()
        for idx, d in results:
            if out is None:
                out = torch.empty(n, d.shape[1], dtype=d.dtype, device=d.device)
            out[torch.tensor(idx, device=d.device)] = d

    with torch.no_grad():
        for i, (img,) in enumerate(tqdm.tqdm(loader, desc, total=n)):
            img = common.variables([img], net.iscuda)[0]
            key = (tuple(img.shape[1:]), img.dtype)
            idx, imgs = buckets.setdefault(key, ([], []))
            idx.append(i)
            imgs.append(img)
            pending += 1
            if len(idx) == batch_size:
                run(key)
            elif pending >= cap:                   # many rare sizes: flush the fullest bucket
                run(max(buckets, key=lambda k: len(buckets[k][0])))
        for key in list(buckets):
            run(key)
        scatter()
    if out is None:                                # empty dataset / shard
        D = net._head_in_dim() if net.without_fc else net.out_dim
        out = torch.empty(0, D, dtype=torch.float32, device='cuda')
    return out


def extract_multiscale_features(dataset, scales, net, desc="Extract feats...", iscuda=True, threads=8):
    """All scales of a multi-scale run from ONE decode and ONE upload per image: the raw uint8
    picture goes to the GPU, each `Scale` (None = original size) is applied there with the
    Pillow-identical resize kernel, and the network runs once per scale.  Returns [N, S*D], the
    per-scale descriptors side by side - the same values as S passes of extract_image_features with
    the chains 'Scale(..)' (test_dir.py:47-94, 118-122), which decode and resize S times on the CPU."""
    from . import ops
    loader = get_loader(dataset, trf_chain='', preprocess=net.preprocess, iscuda=iscuda,
                        output=['img'], batch_size=1, threads=threads, shuffle=False)
    net.eval()
    rows = []
    pool_ = StreamPool()            # the scales of an image (and the next image's) overlap on the device
    with torch.no_grad():
        for (img,) in tqdm.tqdm(loader, desc, total=len(dataset)):
            img = common.variables([img], net.iscuda)[0]              # [1, H, W, 3] uint8
            H, W = int(img.shape[1]), int(img.shape[2])
            per_scale = []
            for sc in scales:
                size = (W, H) if sc is None else sc.target_size((W, H))
                x = img if size == (W, H) else ops.resize_bilinear_u8(img, size)
                per_scale.append(pool_.run(lambda: net(x).reshape(1, -1), x))
            rows.append(per_scale)
        pool_.join()
        rows = [torch.cat(per_scale, dim=1) for per_scale in rows]
    if not rows:
        D = net._head_in_dim() if net.without_fc else net.out_dim
        return torch.empty(0, len(scales) * D, dtype=torch.float32, device='cuda')
    return _check_finite(torch.cat(rows, dim=0), net)


def extract_per_scale(dataset, trfs, net, desc, threads=8, batch_size=16, sharded=False):
    """One [N, D] descriptor tensor per transform chain of `trfs` (a string or a list of strings,
    test_dir.py:118-120).  `sharded`: split the images over the torch.distributed ranks and
    all-gather (dirtorch_amd.distributed); otherwise every rank extracts the whole set."""
    chains = [trfs] if isinstance(trfs, str) else list(trfs)
    run = ddist.extract_sharded if sharded else (lambda fn, ds, arg, net, width=None, **kw: fn(ds, arg, net, **kw))
    scales = None
    if os.environ.get('DIRTORCH_AMD_DEVICE_SCALE', '1') != '0' and net.iscuda:
        scales = trf_mod.device_scales(chains, **net.preprocess)
    if scales is not None and any(s is not None for s in scales):
        # every chain is '' or one Scale(..): decode + upload once, resize per scale on the GPU
        D = net._head_in_dim() if net.without_fc else net.out_dim
        fused = run(extract_multiscale_features, dataset, scales, net, width=len(scales) * D, desc=desc,
                    iscuda=net.iscuda, threads=threads)
        return list(fused.split(D, dim=1))
    return [run(extract_image_features, dataset, chain, net, desc=desc, iscuda=net.iscuda, threads=threads,
                batch_size=batch_size, same_size='Pad' in chain or 'Crop' in chain) for chain in chains]


def _mean_ap(aps, detailed, res):
    """Aggregate per-query APs the way test_dir.py:154-167 does: queries whose AP is -1 (no relevant
    image in that mode) are left out of the mean."""
    if isinstance(aps[0], dict):
        for mode in aps[0]:
            vals = [float(a[mode]) for a in aps]
            if detailed:
                res['APs-' + mode] = vals
            res['mAP-' + mode] = float(np.mean([v for v in vals if v >= 0]))
    else:
        vals = [float(a) for a in aps]
        if detailed:
            res['APs'] = vals
        res['mAP'] = float(np.mean([v for v in vals if v >= 0]))


def eval_model(db, net, trfs, pooling='mean', gemp=3, detailed=False, whiten=None,
               aqe=None, adba=None, threads=8, batch_size=16, save_feats=None,
               load_feats=None, dbg=()):
    """Evaluate `net` on a retrieval dataset that carries its own AP protocol; returns the dict of
    mAP (and top-k, when the dataset has labels) the reference returns."""
    print("\n>> Evaluation...")
    query_db = db.get_query_db()
    same_set = query_db is db

    if load_feats:
        bdescs = np.load(os.path.join(load_feats, 'feats.bdescs.npy'))
        qdescs = bdescs if same_set else np.load(os.path.join(load_feats, 'feats.qdescs.npy'))
    else:
        kw = dict(threads=threads, batch_size=batch_size)
        per_scale_b = extract_per_scale(db, trfs, net, desc="DB", sharded=True, **kw)
        per_scale_q = per_scale_b if same_set else extract_per_scale(query_db, trfs, net, desc="query", **kw)
        bdescs = common.l2_normalize(pool(per_scale_b, pooling, gemp))    # multi-scale pooling, then L2
        qdescs = common.l2_normalize(pool(per_scale_q, pooling, gemp))

    if save_feats:
        mkdir(save_feats)
        np.save(os.path.join(save_feats, 'feats.bdescs.npy'), tonumpy(bdescs))
        if not same_set:
            np.save(os.path.join(save_feats, 'feats.qdescs.npy'), tonumpy(qdescs))

    if whiten is not None:
        bdescs = common.whiten_features(tonumpy(bdescs), net.pca, **whiten)
        qdescs = common.whiten_features(tonumpy(qdescs), net.pca, **whiten)
    # the reference reads a module-global `args` for these two (test_dir.py:141,143); the function
    # parameters are what is meant
    if adba is not None:
        bdescs = expand_descriptors(bdescs, **adba)
    if aqe is not None:
        qdescs = expand_descriptors(qdescs, db=bdescs, **aqe)

    # Large databases (>= 50k images, or DIRTORCH_AMD_DEVICE_RANK=1): the score matrix stays on the
    # GPU and is ranked there instead of being downloaded and argsort-ed row by row.
    flag = os.environ.get('DIRTORCH_AMD_DEVICE_RANK', 'auto')
    on_device = hasattr(db, 'junk') and (flag == '1' or (flag == 'auto' and len(db) >= 50000))
    res = {}
    if on_device:
        from . import ranking
        aps = ranking.eval_aps_device(db, ranking.similarity_device(qdescs, bdescs))
        _mean_ap(aps, detailed, res)
        return res      # the revisitop datasets carry no labels: no top-k (dataset.py:97)

    scores = matmul(qdescs, bdescs)
    try:
        _mean_ap([db.eval_query_AP(q, s) for q, s in enumerate(tqdm.tqdm(scores, desc='AP'))], detailed, res)
    except NotImplementedError:
        print(" AP not implemented!")
    try:
        tops = [db.eval_query_top(q, s) for q, s in enumerate(tqdm.tqdm(scores, desc='top1'))]
        if detailed:
            res['tops'] = tops
        for k in tops[0]:
            res['top%d' % k] = float(np.mean([t[k] for t in tops]))
    except NotImplementedError:
        pass
    return res


def load_model(path, iscuda):
    """Checkpoint file -> network on the GPU, with `preprocess` and the PCA dict it may carry."""
    ck = common.load_checkpoint(path, iscuda)
    net = common.switch_model_to_cuda(nets.create_model(pretrained="", **ck['model_options']), iscuda, ck)
    net.load_state_dict(ck['state_dict'])
    net.preprocess = ck.get('preprocess', net.preprocess)
    if 'pca' in ck:
        net.pca = ck['pca']
    return net


def setup_devices(gpus):
    """--gpu N (the reference's flag, common.torch_set_gpu) for a single process; under
    torch.distributed.run the launcher's LOCAL_RANK picks the GPU and the process group is joined."""
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        ddist.init_from_env()
        return True
    return common.torch_set_gpu([0] if gpus is None else gpus)


# flags shared by test_dir and extract_features: (name, kwargs); same names/defaults as the reference
_COMMON_FLAGS = [
    (('--dataset', '-d'), dict(type=str, required=True, help='Command to load dataset')),
    (('--checkpoint',), dict(type=str, required=True, help='path to weights')),
    (('--trfs',), dict(type=str, required=False, default='', nargs='+', help='test transforms (can be several)')),
    (('--pooling',), dict(type=str, default='gem', help='pooling scheme if several trf chains')),
    (('--gemp',), dict(type=int, default=3, help='GeM pooling power')),
    (('--out-json',), dict(type=str, default='', help='path to output json')),
    (('--detailed',), dict(action='store_true', help='return detailed evaluation')),
    (('--threads',), dict(type=int, default=8, help='number of thread workers')),
    (('--dbg',), dict(default=(), nargs='*', help='debugging options')),
    (('--whitenv',), dict(type=int, default=None, help='number of components, default is None (i.e. all components)')),
    (('--whitenm',), dict(type=float, default=1.0, help='whitening multiplier, default is 1.0 (i.e. no multiplication)')),
]


def build_parser(description='Evaluate a model', extra=()):
    import argparse
    parser = argparse.ArgumentParser(description=description)
    for names, kw in list(_COMMON_FLAGS) + list(extra):
        parser.add_argument(*names, **kw)
    return parser


def select_whitening(net, args):
    """--whiten NAME picks one PCA of the checkpoint's dict (test_dir.py:237-243)."""
    if args.whiten:
        net.pca = net.pca[args.whiten]
        return {'whitenp': args.whitenp, 'whitenv': args.whitenv, 'whitenm': args.whitenm}
    net.pca = None
    return None


def main(argv=None):
    args = build_parser(extra=[
        (('--save-feats',), dict(type=str, default='', help='path to output features')),
        (('--load-feats',), dict(type=str, default='', help='path to load features from')),
        (('--gpu',), dict(type=int, default=0, nargs='+', help='GPU ids')),
        (('--whiten',), dict(type=str, default='Landmarks_clean', help='applies whitening')),
        (('--aqe',), dict(type=int, nargs='+', help='alpha-query expansion paramenters')),
        (('--adba',), dict(type=int, nargs='+', help='alpha-database augmentation paramenters')),
        (('--whitenp',), dict(type=float, default=0.25, help='whitening power, default is 0.5 (i.e., the sqrt)')),
    ]).parse_args(argv)
    iscuda = setup_devices(args.gpu)
    qe = {name: (None if val is None else {'k': val[0], 'alpha': val[1]})
          for name, val in (('aqe', args.aqe), ('adba', args.adba))}

    dataset = datasets.create(args.dataset)
    print("Test dataset:", dataset)
    net = load_model(args.checkpoint, iscuda)
    whiten = select_whitening(net, args)

    res = eval_model(dataset, net, args.trfs, pooling=args.pooling, gemp=args.gemp, detailed=args.detailed,
                     threads=args.threads, dbg=args.dbg, whiten=whiten, aqe=qe['aqe'], adba=qe['adba'],
                     save_feats=args.save_feats, load_feats=args.load_feats)
    if ddist.rank() == 0:
        # (--detailed adds per-query lists; the reference's '%g' print dies on them)
        print(' * ' + '\n * '.join('%s = %g' % kv for kv in res.items() if np.isscalar(kv[1])))
        if args.out_json:
            try:
                merged = json.load(open(args.out_json))
            except IOError:
                merged = {}
            merged[args.dataset] = res
            mkdir(args.out_json, isfile=True)
            with open(args.out_json, 'w') as f:
                f.write(json.dumps(merged, indent=1))
            print("saved to " + args.out_json)
    return res


if __name__ == '__main__':
    main(sys.argv[1:])
