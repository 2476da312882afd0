"""PCA whitening at database scale (SURVEY §8 a8 at configs[3]'s size; dirtorch/utils/common.py:221-239, test_dir.py:136-138).

Sets of >= 32768 rows whose operands are bounded run as two fp16 planes per operand on the matrix cores
(csrc/sim_split.hip whiten_split_kernel, dir_pca_whiten_l2_unit).  Gate: every output within 1.5e-6 of the fp64 product,
relative to the largest entry of its row, AND closer to fp64 than the exact fp32 MFMA chain it replaces (measured on the
MI355X: 1.1-1.2e-6 against 2.8-3.2e-6 for the k-ordered fp32 chain - what is left is the fp32 accumulation of 128 MFMA
steps, not the planes) - on descriptors with a strong common mean (all-positive GeM-like vectors: the case where
subtracting the mean AFTER the product would cancel); 2e-6 on the L2-normalised rows whiten_features returns."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def descriptors(n, D, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = torch.randn(n, D, device='cuda', generator=g).abs_() + 0.25 * torch.randn(n, D, device='cuda', generator=g)
    return torch.nn.functional.normalize(x, dim=1).contiguous()


def pca_params(X, v, seed):
    """mean, orthonormal components [v, D], variances - the attributes common.transform reads (common.py:224-228)."""
    D = X.shape[1]
    mean = X[:8192].mean(dim=0).contiguous()
    g = torch.Generator().manual_seed(seed)
    q, _ = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))
    comps = q[:v].float().contiguous().cuda()
    var = torch.logspace(-1, -5, v)              # explained variances over four decades, as a real PCA's
    return mean, comps, var


def fp64_rows(X, mean, comps, alpha, rows):
    x = X[rows].double().cpu().numpy() - mean.double().cpu().numpy()
    out = x @ comps.double().cpu().numpy().T
    if alpha is not None:
        out = out * alpha.double().cpu().numpy()
    return out


@pytest.mark.parametrize('N,v', [(40013, 2048), (33000, 128), (32768, 100), (100003, 2048)],
                         ids=['40013x2048', '33000x128', '32768x100', '100003x2048'])
def test_split_whitening_vs_fp64(N, v):
    from dirtorch_amd import ops
    D = 2048
    X = descriptors(N, D, 3)
    mean, comps, var = pca_params(X, v, 5)
    alpha = (1.0 / var.double().pow(0.25)).float().cuda()        # whitenp = 0.25, whitenm = 1 (README.md:105-117)
    got = ops.pca_whiten(X, comps, mean, alpha, unit_range=True)
    exact = ops.pca_whiten(X, comps, mean, alpha, unit_range=False)
    assert got.shape == (N, v) and torch.isfinite(got).all()
    rows = np.unique(np.concatenate([np.arange(0, 300), np.arange(N - 300, N), np.linspace(0, N - 1, 1500).astype(np.int64)]))
    ref = fp64_rows(X, mean, comps, alpha, rows)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    e_split = np.abs(got[rows].double().cpu().numpy() - ref) / scale
    e_exact = np.abs(exact[rows].double().cpu().numpy() - ref) / scale
    print('\n[whiten] %d x %d x %d: max |d| / max|row|  two-plane %.2e   exact fp32 chain %.2e' % (N, D, v, e_split.max(), e_exact.max()))
    assert e_split.max() < 1.5e-6 and e_split.max() < e_exact.max(), (e_split.max(), e_exact.max())
    assert not torch.equal(got, exact)                          # (the two-plane kernel did run: another association of the sum)
    # rows the reference did not visit: the two device paths agree everywhere to the sum of their distances from fp64
    d = (got - exact).abs().amax(dim=1) / exact.abs().amax(dim=1)
    assert float(d.max()) < 6e-6, float(d.max())
    # with the row normalisation of whiten_features
    gn = ops.pca_whiten(X, comps, mean, alpha, l2norm=True, unit_range=True)[rows].double().cpu().numpy()
    rn = ref / np.linalg.norm(ref, axis=1, keepdims=True)
    assert np.abs(gn - rn).max() < 2e-6


def test_whiten_features_picks_the_split_path_and_matches_the_oracle():
    """common.whiten_features on a database-sized ndarray: same contract as the reference (ndarray in, ndarray out, dtype by NumPy
    promotion), values within 1e-6 of the oracle's fp64 restatement (oracle/dir_oracle.py whiten_features)."""
    import dir_oracle as O
    from dirtorch_amd.utils import common
    N, D, v = 36000, 512, 96
    X = descriptors(N, D, 9).cpu().numpy()
    P = O.fit_pca(X[:4096].astype(np.float64))
    P = O.PCAParams(P.mean_.astype(np.float32), P.components_.astype(np.float32), P.explained_variance_.astype(np.float32))
    got = common.whiten_features(X, P, whitenp=0.25, whitenv=v)
    assert got.dtype == np.float32 and got.shape == (N, v)
    P64 = O.PCAParams(P.mean_.astype(np.float64), P.components_.astype(np.float64), P.explained_variance_.astype(np.float64))
    rows = np.linspace(0, N - 1, 3000).astype(np.int64)
    ref = O.whiten_features(X[rows].astype(np.float64), P64, whitenp=0.25, whitenv=v)
    assert np.abs(got[rows] - ref).max() < 2e-6


def test_out_of_range_operands_come_out_non_finite():
    """The two-plane form is the caller's promise (|X - mean| < 64): a value beyond it must show, not pass silently."""
    from dirtorch_amd import ops
    N, D, v = 32768, 256, 64
    X = descriptors(N, D, 1)
    mean, comps, _ = pca_params(X, v, 2)
    X[777, 5] = 100.0
    out = ops.pca_whiten(X, comps, mean, None, unit_range=True)
    assert not torch.isfinite(out[777]).all()
    assert torch.isfinite(out[:777]).all() and torch.isfinite(out[778:]).all()
    assert torch.isfinite(ops.pca_whiten(X, comps, mean, None, unit_range=False)).all()
