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
    scores = ops.gemm_nt(_dev(B), _dev(A))
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
    out = ops.gemm_nt(_dev(comps), _dev(X),
                      qsub=None if mean is None else _dev(np.asarray(mean).reshape(-1)),
                      alpha=None if alpha is None else _dev(alpha.astype(np.float32)))
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


# ---- runtime (common.py:58-218) ---------------------------------------------------------------
def torch_set_gpu(gpus, seed=None, randomize=True):
    if type(gpus) is int:
        gpus = [gpus]
    assert gpus, 'error: empty gpu list, use --gpu N N ...'
    cuda = all(gpu >= 0 for gpu in gpus)
    if cuda:
        if any(gpu >= 1000 for gpu in gpus):
            visible_gpus = [int(gpu) for gpu in os.environ['CUDA_VISIBLE_DEVICES'].split(',')]
            os.environ['CUDA_VISIBLE_DEVICES'] = ','.join([str(visible_gpus[gpu - 1000]) for gpu in gpus])
        else:
            os.environ['CUDA_VISIBLE_DEVICES'] = ','.join([str(gpu) for gpu in gpus])
        assert cuda and torch.cuda.is_available(), "%s has GPUs %s unavailable" % (
            os.environ.get('HOSTNAME', '?'), os.environ['CUDA_VISIBLE_DEVICES'])
        print('Launching on GPUs ' + os.environ['CUDA_VISIBLE_DEVICES'])
    else:
        # the reference falls back to CPU here; this engine is MI355X-only by design
        raise RuntimeError('dirtorch_amd has no CPU execution path: pass --gpu with a device id')
    torch_set_seed(seed, cuda, randomize=randomize)
    return cuda


def torch_set_seed(seed, cuda, randomize=True):
    if randomize and not seed:
        import time
        try:
            seed = int(np.uint32(hash(time.time())))
        except OverflowError:
            seed = int.from_bytes(os.urandom(4), byteorder='little', signed=False)
    if seed:
        np.random.seed(seed)
        torch.manual_seed(seed)
        if cuda:
            torch.cuda.manual_seed(seed)


def torch_load_trusted(filename):
    """torch.load for the reference's checkpoint format.  Real checkpoints pickle an
    sklearn.decomposition.PCA under 'pca' (test_dir.py:189-190), which torch >= 2.6 refuses under
    weights_only=True - the checkpoint is a trusted local file, as in the reference (common.py:121)."""
    return torch.load(filename, map_location=lambda storage, loc: storage, weights_only=False)


def save_checkpoint(state, is_best, filename):
    import shutil
    dirs = os.path.split(filename)[0]
    if dirs and not os.path.isdir(dirs):
        os.makedirs(dirs)
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, filename + '.best')
        filename = filename + '.best'
    print("saving to " + filename)


def load_checkpoint(filename, iscuda=False):
    if not filename:
        return None
    assert os.path.isfile(filename), "=> no checkpoint found at '%s'" % filename
    checkpoint = torch_load_trusted(filename)
    print("=> loading checkpoint '%s'" % filename, end='')
    for key in ['epoch', 'iter', 'current_iter']:
        if key in checkpoint:
            print(" (%s %d)" % (key, checkpoint[key]), end='')
    print()

    new_dict = OrderedDict()
    for k, v in list(checkpoint['state_dict'].items()):
        if k.startswith('module.'):
            k = k[7:]
        new_dict[k] = v
    checkpoint['state_dict'] = new_dict
    return checkpoint


def switch_model_to_cuda(model, iscuda=True, checkpoint=None):
    """The reference wraps the model in nn.DataParallel here (common.py:150-175); this engine is
    one process per GPU, so the model is simply placed on the current device.  Checkpoint keys are
    left without the 'module.' prefix (load_state_dict accepts both)."""
    if iscuda:
        try:
            model.cuda()
            model.isasync = True
        except RuntimeError as e:
            print("RuntimeError:", e, "(machine %s, GPU %s)" % (
                os.environ.get('HOSTNAME', '?'), os.environ.get('CUDA_VISIBLE_DEVICES', '?')),
                file=sys.stderr)
            sys.exit(1)
    else:
        raise RuntimeError('dirtorch_amd has no CPU execution path (iscuda=False)')
    model.iscuda = iscuda
    return model


def model_size(model):
    size = 0
    for weights in model.state_dict().values():
        size += np.prod(weights.shape)
    return size


def variables(inputs, iscuda, not_on_gpu=[]):
    """Move a list of tensors to the GPU (common.py:205-218)."""
    inputs_var = []
    for i, x in enumerate(inputs):
        if i not in not_on_gpu and not isinstance(x, (tuple, list)):
            if iscuda:
                x = x.cuda(non_blocking=True)
        inputs_var.append(x)
    return inputs_var
