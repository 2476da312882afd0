"""Post-processing + runtime helpers - same names and argument meaning as dirtorch/utils/common.py.

    pool(list, pooling, gemp)                       common.py:41-55    -> HIP multiscale_pool kernel
    whiten_features(X, pca, l2norm, whitenp, ...)   common.py:221-239  -> fp32 MFMA GEMM + L2 kernels
    matmul(A, B) -> ndarray                         common.py:30-38    -> fp32 MFMA GEMM kernel
    tonumpy, variables, torch_set_gpu, torch_set_seed, load_checkpoint, switch_model_to_cuda
Nothing here computes on the CPU: host inputs are uploaded, processed by the engine's kernels and
downloaded, exactly where the reference crosses the device boundary (common.py:25,35).
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

from .. import ops


def typename(x):
    return type(x).__module__


def tonumpy(x):
    if typename(x) == torch.__name__:
        return x.cpu().numpy()
    else:
        return x


def _dev(x):
    """float32 contiguous CUDA tensor from an ndarray / tensor."""
    if typename(x) == np.__name__:
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return x.to(device='cuda', dtype=torch.float32).contiguous()


def matmul(A, B):
    """Similarity scores A.B^T as a NumPy array [len(A), len(B)] (common.py:30-38)."""
    if typename(A) == np.__name__:
        B = tonumpy(B)
    elif typename(B) != torch.__name__:
        raise TypeError("matrices must be either numpy or torch type")
    # P = database rows (long operand, read once from HBM), Q = queries
    scores = ops.similarity(_dev(A), _dev(B))
    return scores.cpu().numpy()


def pool(x, pooling='mean', gemp=3):
    """Combine descriptors of several scales (common.py:41-55): list of [N,D] -> [N,D]."""
    if len(x) == 1:
        return x[0]
    if pooling not in ('mean', 'gem'):
        raise ValueError("Bad pooling mode: " + str(pooling))
    dev = x[0].device
    xs = torch.stack([t.to(device='cuda', dtype=torch.float32) for t in x], dim=0).contiguous()
    return ops.multiscale_pool(xs, pooling, gemp).to(dev)


def l2_normalize(x, eps=1e-12):
    """Row-wise F.normalize(x, p=2, dim=1) on the engine (test_dir.py:121-122)."""
    dev = x.device
    y = x.to(device='cuda', dtype=torch.float32).clone().contiguous()
    return ops.l2norm_rows_(y, eps).to(dev)


WHITEN_SPLIT_MIN_ROWS = 32768     # include/dir_engine.h dir_pca_whiten_l2_unit: below this the exact fp32 chain runs anyway


def _transform_dev(pca, X, whitenp, whitenv, whitenm, use_sklearn):
    """Device-side PCA projection; returns (CUDA tensor [N,v], dtype NumPy promotion would give)."""
    res_dtype = np.float32
    if use_sklearn:
        comps = np.asarray(pca.components_[:whitenv])
        mean = pca.mean_
        alpha = None
        if pca.whiten:
            alpha = 1.0 / (whitenm * np.power(np.asarray(pca.explained_variance_[:whitenv], dtype=np.float64), whitenp))
        res_dtype = np.result_type(np.asarray(X).dtype, comps.dtype)
    else:
        comps = np.asarray(pca['W']).T
        mean = pca['means']
        alpha = None
        res_dtype = np.result_type(np.asarray(X).dtype, comps.dtype)
    Xd, Cd = _dev(X), _dev(comps)
    md = None if mean is None else _dev(np.asarray(mean).reshape(-1))
    ad = None if alpha is None else _dev(alpha.astype(np.float32))
    if Xd.shape[0] >= WHITEN_SPLIT_MIN_ROWS and Xd.shape[1] % 32 == 0:
        # a database-sized set (test_dir.py:136-138 at the 10^6-distractor protocol): when X - mean and the components are
        # bounded (L2-normalised descriptors, PCA rows) the product runs on two fp16 planes per operand (csrc/sim_split.hip
        # whiten_split_kernel, ~4x the exact fp32 chain, ~1e-6 of the fp64 result); one abs-max pass over X decides
        from .. import ranking
        bound = ranking.UNIT_RANGE_BOUND - (float(md.abs().max()) if md is not None else 0.0)
        lo, hi = torch.aminmax(Xd)
        if max(-float(lo), float(hi)) < bound and ranking.is_unit_range(Cd):
            return ops.pca_whiten(Xd, Cd, md, ad, unit_range=True), res_dtype
    out = ops.gemm_nt(Cd, Xd, qsub=md, alpha=ad)
    return out, res_dtype


def transform(pca, X, whitenp=0.5, whitenv=None, whitenm=1.0, use_sklearn=True):
    """PCA projection (+ variance rescaling) of X [N,D] (common.py:221-232); returns ndarray."""
    res, res_dtype = _transform_dev(pca, tonumpy(X), whitenp, whitenv, whitenm, use_sklearn)
    return res.cpu().numpy().astype(res_dtype, copy=False)


def whiten_features(X, pca, l2norm=True, whitenp=0.5, whitenv=None, whitenm=1.0, use_sklearn=True):
    """PCA-whiten descriptors and L2-normalise the rows (common.py:235-239); ndarray in, ndarray out."""
    X = tonumpy(X)
    res, res_dtype = _transform_dev(pca, X, whitenp, whitenv, whitenm, use_sklearn)
    if l2norm:
        # np.linalg.norm has no eps: a zero row gives NaN in the reference; eps=0 reproduces that
        ops.l2norm_rows_(res, 0.0)
    return res.cpu().numpy().astype(res_dtype, copy=False)


# ---- runtime helpers (device selection, seeding, checkpoints; common.py:58-218) -----------------
def torch_set_gpu(gpus, seed=None, randomize=True):
    """Select the GPU(s) through CUDA_VISIBLE_DEVICES (honoured by PyTorch-ROCm) and seed the RNGs.
    Ids >= 1000 index into the already-visible list, as in the reference.  Returns True; a negative
    id (the reference's "run on CPU") is an error here: the engine is MI355X-only."""
    ids = [gpus] if isinstance(gpus, int) else list(gpus)
    assert ids, 'error: empty gpu list, use --gpu N N ...'
    if any(g < 0 for g in ids):
        raise RuntimeError('dirtorch_amd has no CPU execution path: pass --gpu with a device id')
    if any(g >= 1000 for g in ids):
        visible = [int(v) for v in os.environ['CUDA_VISIBLE_DEVICES'].split(',')]
        ids = [visible[g - 1000] for g in ids]
    os.environ['CUDA_VISIBLE_DEVICES'] = ','.join(str(g) for g in ids)
    assert torch.cuda.is_available(), "%s has GPUs %s unavailable" % (
        os.environ.get('HOSTNAME', '?'), os.environ['CUDA_VISIBLE_DEVICES'])
    print('Launching on GPUs ' + os.environ['CUDA_VISIBLE_DEVICES'])
    torch_set_seed(seed, True, randomize=randomize)
    return True


def torch_set_seed(seed, cuda, randomize=True):
    """Seed numpy / torch (/ the GPU generator); with no seed and randomize=True draw one."""
    if not seed and randomize:
        seed = int.from_bytes(os.urandom(4), byteorder='little', signed=False)
    if not seed:
        return
    np.random.seed(seed % (1 << 32))
    torch.manual_seed(seed)
    if cuda:
        torch.cuda.manual_seed(seed)


def torch_load_trusted(filename):
    """torch.load for the reference's checkpoint format.  Real checkpoints pickle an
    sklearn.decomposition.PCA under 'pca' (test_dir.py:189-190), which torch >= 2.6 refuses under
    weights_only=True - the checkpoint is a trusted local file, as in the reference (common.py:121)."""
    return torch.load(filename, map_location='cpu', weights_only=False)


def save_checkpoint(state, is_best, filename):
    """torch.save with the reference's '.best' copy (common.py:102-114)."""
    folder = os.path.dirname(filename)
    if folder:
        os.makedirs(folder, exist_ok=True)
    torch.save(state, filename)
    if is_best:
        import shutil
        shutil.copyfile(filename, filename + '.best')
    print("saving to " + (filename + '.best' if is_best else filename))


def load_checkpoint(filename, iscuda=False):
    """Read a checkpoint dict {'model_options', 'state_dict', ['preprocess'], ['pca'], ...} and drop
    the DataParallel 'module.' prefix from its parameter names (common.py:117-147)."""
    if not filename:
        return None
    assert os.path.isfile(filename), "=> no checkpoint found at '%s'" % filename
    ck = torch_load_trusted(filename)
    progress = ''.join(" (%s %d)" % (k, ck[k]) for k in ('epoch', 'iter', 'current_iter') if k in ck)
    print("=> loading checkpoint '%s'%s" % (filename, progress))
    ck['state_dict'] = OrderedDict((k[7:] if k.startswith('module.') else k, v)
                                   for k, v in ck['state_dict'].items())
    return ck


def switch_model_to_cuda(model, iscuda=True, checkpoint=None):
    """Place the model on the current GPU.  The reference wraps it in nn.DataParallel here
    (common.py:150-175) and re-prefixes the checkpoint keys; this engine runs one process per GPU
    (dirtorch_amd.distributed shards the work), so keys stay un-prefixed - load_state_dict takes both."""
    if not iscuda:
        raise RuntimeError('dirtorch_amd has no CPU execution path (iscuda=False)')
    try:
        model.cuda()
    except RuntimeError as e:
        print("RuntimeError:", e, "(machine %s, GPU %s)" % (
            os.environ.get('HOSTNAME', '?'), os.environ.get('CUDA_VISIBLE_DEVICES', '?')), file=sys.stderr)
        sys.exit(1)
    model.isasync = True
    model.iscuda = True
    return model


def model_size(model):
    """Number of scalars in the state dict."""
    return int(sum(int(np.prod(t.shape)) for t in model.state_dict().values()))


def variables(inputs, iscuda, not_on_gpu=()):
    """Move the tensors of `inputs` to the GPU (non-blocking), except indices in `not_on_gpu` and
    nested lists (common.py:205-218; Variable wrapping is a no-op in today's torch)."""
    return [x.cuda(non_blocking=True)
            if iscuda and i not in not_on_gpu and not isinstance(x, (tuple, list)) else x
            for i, x in enumerate(inputs)]
