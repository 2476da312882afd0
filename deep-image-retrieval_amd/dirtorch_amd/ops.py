"""Tensor-level wrappers of the per-op C-ABI entry points (include/dir_engine.h).

torch is used for device memory and the current stream only; every computation below is a HIP
kernel of libdir_engine.so.  16-bit activations travel as torch.bfloat16 / torch.float16 tensors in
NHWC layout.
"""
import ctypes

import torch

from . import _lib
from ._lib import DIR_BF16, DIR_FP16, POOLING, call, ptr, stream_ptr


def _dtype_code(t):
    if t.dtype == torch.bfloat16:
        return DIR_BF16
    if t.dtype == torch.float16:
        return DIR_FP16
    raise TypeError('16-bit activation tensor expected (bfloat16 or float16), got %s' % t.dtype)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError('device tensor expected')
        if t is not None and not t.is_contiguous():
            raise ValueError('contiguous tensor expected')


def conv_variant_names():
    lib = _lib.load()
    out = []
    for v in range(lib.dir_conv_variant_count()):
        buf = ctypes.create_string_buffer(64)
        call('dir_conv_variant_name', v, buf, 64)
        out.append(buf.value.decode())
    return out


def pack_conv_weight(w_oihw, dtype=torch.bfloat16):
    """OIHW fp32 -> [Cout][R][S][Cin] 16-bit (what dir_conv_bn_act expects)."""
    return w_oihw.permute(0, 2, 3, 1).contiguous().to(dtype)


def pack_stem_weight(w_oihw, dtype=torch.bfloat16):
    """7x7 s2 stem weight [64,3,7,7] -> 4x4 s1 space-to-depth form [64][4][4][16]
    (tap r = 2R + dy - 1, s = 2S + dx - 1, channel (dy*2+dx)*3 + c); mirrors engine.hip."""
    O = w_oihw.shape[0]
    out = torch.zeros(O, 4, 4, 16, dtype=torch.float32, device=w_oihw.device)
    for R in range(4):
        for S in range(4):
            for dy in range(2):
                for dx in range(2):
                    r, s = 2 * R + dy - 1, 2 * S + dx - 1
                    if 0 <= r < 7 and 0 <= s < 7:
                        c0 = (dy * 2 + dx) * 3
                        out[:, R, S, c0:c0 + 3] = w_oihw[:, :, r, s]
    return out.to(dtype)


def conv_bn_act_f32(x, w, bias, res=None, stride=1, pad=0, relu=True):
    """The strict path's convolution (dir_conv_bn_act_f32, conv_f32.hip): x NHWC [B,H,W,Cin] fp32,
    w [Cout,R,S,Cin] fp32, bias fp32 [Cout], res NHWC fp32 or None -> NHWC [B,OH,OW,Cout] fp32."""
    _need_cuda(x, w, bias, res)
    for t in (x, w, bias, res):
        if t is not None and t.dtype != torch.float32:
            raise TypeError('fp32 tensors expected')
    B, H, W, Cin = x.shape
    Cout, R, S, Cin2 = w.shape
    if Cin2 != Cin:
        raise ValueError('Cin mismatch')
    OH = (H + 2 * pad - R) // stride + 1
    OW = (W + 2 * pad - S) // stride + 1
    y = torch.empty(B, OH, OW, Cout, dtype=torch.float32, device=x.device)
    call('dir_conv_bn_act_f32', ptr(x), ptr(w), ptr(bias), ptr(res), ptr(y), B, H, W, Cin, Cout, R, S, stride, pad,
         OH, OW, int(bool(relu)), stream_ptr())
    return y


def conv_bn_act(x, w, bias, res=None, stride=1, pad=0, relu=True, out_hw=None, variant=-1,
                naive=False, ksplit=None):
    """x NHWC [B,H,W,Cin] 16-bit, w [Cout,R,S,Cin] 16-bit, bias fp32 [Cout] -> NHWC [B,OH,OW,Cout].
    ksplit: None = plain launch; n > 1 = split-K into n slices; -1 = the engine's own choice
    (conv_bn_act.last_ksplit holds what was used)."""
    _need_cuda(x, w, bias, res)
    B, H, W, Cin = x.shape
    Cout, R, S, Cin2 = w.shape
    if Cin2 != Cin:
        raise ValueError('Cin mismatch')
    if out_hw is None:
        OH = (H + 2 * pad - R) // stride + 1
        OW = (W + 2 * pad - S) // stride + 1
    else:
        OH, OW = out_hw
    y = torch.empty(B, OH, OW, Cout, dtype=x.dtype, device=x.device)
    args = [ptr(x), ptr(w), ptr(bias), ptr(res), ptr(y), B, H, W, Cin, Cout, R, S, stride, pad, OH,
            OW, int(bool(relu)), _dtype_code(x)]
    if ksplit is not None:
        n = 8 if ksplit < 0 else max(int(ksplit), 1)
        scratch = torch.empty(n * B * OH * OW * Cout, dtype=torch.float32, device=x.device)
        used = ctypes.c_int()
        call('dir_conv_bn_act_splitk', *args, int(variant), int(ksplit), ptr(scratch), scratch.numel() * 4,
             ctypes.byref(used), stream_ptr())
        conv_bn_act.last_ksplit = used.value
    elif naive:
        call('dir_conv_bn_act_naive', *args, stream_ptr())
    else:
        call('dir_conv_bn_act', *args, int(variant), stream_ptr())
    return y


def conv_c3c1(t2, w3, bias3, res, w1, bias1, relu3=True, relu1=True):
    """Fused bottleneck seam (dir_conv_c3c1): returns (y [B,H,W,4P], t1 [B,H,W,P2]).
    t2 [B,H,W,P], w3 [4P,1,1,P] or [4P,P], res [B,H,W,4P], w1 [P2,1,1,4P] or [P2,4P]; 16-bit NHWC, fp32 biases."""
    _need_cuda(t2, w3, bias3, res, w1, bias1)
    B, H, W, P = t2.shape
    P2 = w1.shape[0]
    y = torch.empty(B, H, W, 4 * P, dtype=t2.dtype, device=t2.device)
    t1 = torch.empty(B, H, W, P2, dtype=t2.dtype, device=t2.device)
    call('dir_conv_c3c1', ptr(t2), ptr(w3), ptr(bias3), ptr(res), ptr(y), ptr(w1), ptr(bias1), ptr(t1), B, H, W, P, P2,
         int(bool(relu3)), int(bool(relu1)), _dtype_code(t2), stream_ptr())
    return y, t1


def conv_c3c1_ds(t2, x, wcat, bias, w1, bias1, relu3=True, relu1=True):
    """Fused seam with the downsample branch folded in as extra K (dir_conv_c3c1_ds): t2, x [B,H,W,64],
    wcat [256, 128] = [w3 | wds], bias = bias3 + bias_ds -> (y [B,H,W,256], t1 [B,H,W,64])."""
    _need_cuda(t2, x, wcat, bias, w1, bias1)
    B, H, W, P = t2.shape
    y = torch.empty(B, H, W, 256, dtype=t2.dtype, device=t2.device)
    t1 = torch.empty(B, H, W, 64, dtype=t2.dtype, device=t2.device)
    call('dir_conv_c3c1_ds', ptr(t2), ptr(x), ptr(wcat), ptr(bias), ptr(y), ptr(w1), ptr(bias1), ptr(t1), B, H, W,
         int(bool(relu3)), int(bool(relu1)), _dtype_code(t2), stream_ptr())
    return y, t1


def conv_c3c1_wpair(t2, w3, bias3, res, w1, bias1, relu3=True, relu1=True):
    """The fused seam with paired weights (dir_conv_c3c1_wpair, fp16, planes 64): w3 = (hi, lo) [256, 64], w1 = (hi, lo)
    or (hi, None) [P2, 256]; t2 [B,H,W,64], res [B,H,W,256] single fp16 planes -> (y [B,H,W,256], t1 [B,H,W,P2])."""
    (w3h, w3l), (w1h, w1l) = w3, w1
    _need_cuda(t2, w3h, w3l, bias3, res, w1h, bias1)
    B, H, W, P = t2.shape
    P2 = w1h.shape[0]
    y = torch.empty(B, H, W, 4 * P, dtype=t2.dtype, device=t2.device)
    t1 = torch.empty(B, H, W, P2, dtype=t2.dtype, device=t2.device)
    call('dir_conv_c3c1_wpair', ptr(t2), ptr(w3h), ptr(w3l), ptr(bias3), ptr(res), ptr(y), ptr(w1h),
         ptr(w1l) if w1l is not None else None, ptr(bias1), ptr(t1), B, H, W, P2, int(bool(relu3)), int(bool(relu1)), stream_ptr())
    return y, t1


def conv_c3c1_ds_wpair(t2, x, wcat, bias, w1, bias1, relu3=True, relu1=True):
    """... and its downsample form (dir_conv_c3c1_ds_wpair): x = (hi, lo) [B,H,W,64] the paired block input,
    wcat = (hi, lo) [256, 128] = [w3 | wds], w1 = (hi, lo) [64, 256] -> (y [B,H,W,256], t1 [B,H,W,64])."""
    (xh, xl), (wh, wl), (w1h, w1l) = x, wcat, w1
    _need_cuda(t2, xh, xl, wh, wl, bias, w1h, w1l, bias1)
    B, H, W, P = t2.shape
    y = torch.empty(B, H, W, 256, dtype=t2.dtype, device=t2.device)
    t1 = torch.empty(B, H, W, 64, dtype=t2.dtype, device=t2.device)
    call('dir_conv_c3c1_ds_wpair', ptr(t2), ptr(xh), ptr(xl), ptr(wh), ptr(wl), ptr(bias), ptr(y), ptr(w1h), ptr(w1l),
         ptr(bias1), ptr(t1), B, H, W, int(bool(relu3)), int(bool(relu1)), stream_ptr())
    return y, t1


def conv_dual(t2, x, wcat, bias, stride2=2, relu=True):
    """conv3 + downsample as one two-source GEMM (dir_conv_dual): t2 [B,OH,OW,Cin], x [B,H2,W2,Cin2],
    wcat [Cout, Cin + Cin2], bias fp32 [Cout] -> y [B,OH,OW,Cout]."""
    _need_cuda(t2, x, wcat, bias)
    B, OH, OW, Cin = t2.shape
    _, H2, W2, Cin2 = x.shape
    Cout = wcat.shape[0]
    y = torch.empty(B, OH, OW, Cout, dtype=t2.dtype, device=t2.device)
    call('dir_conv_dual', ptr(t2), ptr(x), ptr(wcat), ptr(bias), ptr(y), B, OH, OW, Cin, Cout, Cin2, H2, W2,
         int(stride2), int(bool(relu)), _dtype_code(t2), stream_ptr())
    return y


def prep_input(img, dtype=torch.bfloat16, mean=None, std=None):
    """fp32 NCHW (normalised) or uint8 NHWC image batch -> space-to-depth NHWC16 stem input."""
    _need_cuda(img)
    if img.dtype == torch.float32:
        B, C, H, W = img.shape
        fmt = _lib.DIR_IMG_F32_NCHW
    elif img.dtype == torch.uint8:
        B, H, W, C = img.shape
        fmt = _lib.DIR_IMG_U8_NHWC
    else:
        raise TypeError('image must be float32 NCHW or uint8 NHWC')
    if C != 3:
        raise ValueError('3-channel image expected')
    out = torch.empty(B, (H + 1) // 2, (W + 1) // 2, 16, dtype=dtype, device=img.device)
    m = (ctypes.c_float * 3)(*(mean or (0, 0, 0)))
    s = (ctypes.c_float * 3)(*(std or (1, 1, 1)))
    call('dir_prep_input', ptr(img), fmt, m, s, ptr(out), B, H, W,
         DIR_BF16 if dtype == torch.bfloat16 else DIR_FP16, stream_ptr())
    return out


def maxpool_3x3s2(x):
    _need_cuda(x)
    B, H, W, C = x.shape
    y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C, dtype=x.dtype, device=x.device)
    call('dir_maxpool_3x3s2', ptr(x), ptr(y), B, H, W, C, _dtype_code(x), stream_ptr())
    return y


def global_pool(x, pooling='gem', p=3.0, eps=1e-6, center_bias=0.0):
    _need_cuda(x)
    B, H, W, C = x.shape
    out = torch.empty(B, C, dtype=torch.float32, device=x.device)
    call('dir_global_pool', ptr(x), ptr(out), B, H, W, C, POOLING[pooling], float(p), float(eps),
         float(center_bias), _dtype_code(x), stream_ptr())
    return out


def resize_bilinear_u8(img, size):
    """PIL `img.resize((ow, oh), Image.BILINEAR)` of uint8 images on the GPU, bit-identical to Pillow.
    img: [H,W,3] or [B,H,W,3] uint8; size = (ow, oh) like PIL.  The `Scale` transform of
    dirtorch/utils/transforms.py:133-185 without the CPU."""
    _need_cuda(img)
    if img.dtype != torch.uint8 or img.shape[-1] != 3 or img.dim() not in (3, 4):
        raise TypeError('uint8 [H,W,3] or [B,H,W,3] image expected')
    x = img.contiguous().view((-1,) + tuple(img.shape[-3:]))
    B, H, W, _ = x.shape
    ow, oh = int(size[0]), int(size[1])
    need = ctypes.c_size_t()
    call('dir_resize_workspace_bytes', B, H, W, oh, ow, ctypes.byref(need))
    ws = torch.empty(need.value, dtype=torch.uint8, device=x.device)
    out = torch.empty(B, oh, ow, 3, dtype=torch.uint8, device=x.device)
    call('dir_resize_bilinear_u8', ptr(x), ptr(out), B, H, W, oh, ow, ptr(ws), ws.numel(), stream_ptr())
    return out if img.dim() == 4 else out[0]


def upsample_add(x, low):
    """x + nearest-upsampled low: NHWC 16-bit [B,H,W,C] and [B,h,w,C] (rmac_resnet_fpn.py:55-60)."""
    _need_cuda(x, low)
    if x.dtype != low.dtype or x.shape[0] != low.shape[0] or x.shape[3] != low.shape[3]:
        raise ValueError('upsample_add: x and low must share dtype, batch and channels')
    x, low = x.contiguous(), low.contiguous()
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    call('dir_upsample_add', ptr(x), ptr(low), ptr(y), B, H, W, low.shape[1], low.shape[2], C,
         _dtype_code(x), stream_ptr())
    return y


def l2norm_rows_(x, eps=1e-12):
    """In place: x[i] /= max(||x[i]||, eps)."""
    _need_cuda(x)
    if x.dtype != torch.float32 or x.dim() != 2:
        raise TypeError('2-D float32 tensor expected')
    call('dir_l2norm_rows', ptr(x), x.shape[0], x.shape[1], float(eps), stream_ptr())
    return x


def gemm_nt(P, Q, qsub=None, bias=None, alpha=None):
    """out[j][i] = alpha[i] * sum_k P[i][k] * (Q[j][k] - qsub[k]) + bias[i]; fp32, exact MFMA."""
    _need_cuda(P, Q, qsub, bias, alpha)
    for t in (P, Q, qsub, bias, alpha):
        if t is not None and t.dtype != torch.float32:
            raise TypeError('float32 tensors expected')
    NP, K = P.shape
    NQ, K2 = Q.shape
    if K != K2:
        raise ValueError('inner dimensions differ')
    out = torch.empty(NQ, NP, dtype=torch.float32, device=P.device)
    if NP == 0 or NQ == 0:
        return out
    call('dir_gemm_nt_f32', ptr(P), K, ptr(Q), K, ptr(out), NP, NP, NQ, K, ptr(qsub), ptr(bias),
         ptr(alpha), stream_ptr())
    return out


def similarity(queries, database, unit_range=False):
    """scores [Q, N] = queries . database^T, fp32 (dir_similarity).  Databases of >= 32768 rows with a width that
    is a multiple of 32 run as a three-plane bf16 split on the matrix cores (products to 2^-23, fp32
    accumulation; csrc/sim_split.hip), everything else - and everything under DIRTORCH_AMD_SIM_EXACT=1 - as
    the k-ordered fp32 MFMA chain of gemm_nt.
    unit_range=True (dir_similarity_unit): the caller guarantees |values| < 64 (L2-normalised descriptors) - two fp16
    planes and half the matrix work on the large-database path; a larger value gives NON-FINITE scores, not wrong ones."""
    _need_cuda(queries, database)
    for t in (queries, database):
        if t.dtype != torch.float32 or t.dim() != 2 or not t.is_contiguous():
            raise TypeError('contiguous 2-D float32 tensors expected')
    Q, D = queries.shape
    N, D2 = database.shape
    if D != D2:
        raise ValueError('inner dimensions differ')
    out = torch.empty(Q, N, dtype=torch.float32, device=database.device)
    if Q == 0 or N == 0:
        return out
    call('dir_similarity_unit' if unit_range else 'dir_similarity', ptr(queries), Q, ptr(database), N, D, ptr(out), stream_ptr())
    return out


def multiscale_pool(xs, pooling='mean', gemp=3.0):
    """xs: [S,N,D] fp32 -> [N,D] (mean or signed-power mean); no final L2."""
    _need_cuda(xs)
    S, N, D = xs.shape
    mode = {'mean': 0, 'gem': 1}[pooling]
    out = torch.empty(N, D, dtype=torch.float32, device=xs.device)
    call('dir_multiscale_pool', ptr(xs), ptr(out), S, N, D, mode, float(gemp), stream_ptr())
    return out


def rank_counts(scores, probe_idx):
    """scores [Q,N] fp32 (CUDA), probe_idx [Q,P] int32 (-1 = unused) -> (counts [Q,P] int32,
    probe_scores [Q,P] fp32): how many items rank before each probe (np.argsort(...)[::-1] order)."""
    _need_cuda(scores, probe_idx)
    if scores.dtype != torch.float32 or probe_idx.dtype != torch.int32:
        raise TypeError('float32 scores and int32 probe indices expected')
    Q, N = scores.shape
    P = probe_idx.shape[1]
    counts = torch.empty(Q, P, dtype=torch.int32, device=scores.device)    # zeroed by the call
    pscores = torch.zeros(Q, P, dtype=torch.float32, device=scores.device)
    call('dir_rank_counts', ptr(scores), N, Q, N, ptr(probe_idx), P, ptr(counts), ptr(pscores), stream_ptr())
    return counts, pscores


def revisitop_ap(probe_idx, counts, pscores, pos_off, pos_list, junk_off, junk_list, modes):
    """APs of every (query, mode) from the outputs of rank_counts (dir_revisitop_ap): returns a CUDA
    float64 tensor [Q, modes], -1 where a mode has no positive.  pos_* / junk_* are CSR lists of
    positions into the rows of probe_idx (int32 CUDA tensors)."""
    _need_cuda(probe_idx, counts, pscores, pos_off, pos_list, junk_off, junk_list)
    Q, P = probe_idx.shape
    ap = torch.empty(Q, modes, dtype=torch.float64, device=probe_idx.device)
    terms = torch.empty(max(int(pos_list.numel()), 1), dtype=torch.float64, device=probe_idx.device)
    call('dir_revisitop_ap', ptr(probe_idx), Q, P, ptr(counts), ptr(pscores), ptr(pos_off), ptr(pos_list),
         ptr(junk_off), ptr(junk_list), int(modes), ptr(terms), ptr(ap), stream_ptr())
    return ap


def expand_descriptors(descs, db=None, alpha=0.0, k=0, scratch_bytes=256 << 20):
    """alpha-QE / DBA on the device (dir_expand_descriptors): descs [n,D], db [m,D] fp32 CUDA (db None =
    expand the set against itself, a row never being its own neighbour) -> [n,D] fp32 CUDA."""
    _need_cuda(descs, db)
    self_set = db is None
    pool_ = descs if self_set else db
    n, D = descs.shape
    m = pool_.shape[0]
    out = torch.empty_like(descs)
    if n == 0:
        return out
    rows = max(1, min(n, scratch_bytes // max(4 * m, 1)))
    sim = torch.empty(rows * m, dtype=torch.float32, device=descs.device)
    call('dir_expand_descriptors', ptr(descs), n, ptr(pool_), m, D, int(k), float(alpha), int(self_set),
         ptr(out), ptr(sim), sim.numel() * 4, stream_ptr())
    return out


def stem_pool(s2d, w_packed, bias, out_hw):
    """Fused stem: s2d image [B,H2,W2,16] + packed 4x4x16 filter -> pooled [B,PH,PW,64];
    out_hw = (OH, OW) of the 7x7 s2 convolution."""
    _need_cuda(s2d, w_packed, bias)
    B, H2, W2, _ = s2d.shape
    OH, OW = out_hw
    y = torch.empty(B, (OH - 1) // 2 + 1, (OW - 1) // 2 + 1, 64, dtype=s2d.dtype, device=s2d.device)
    call('dir_stem_pool', ptr(s2d), ptr(w_packed), ptr(bias), ptr(y), B, H2, W2, OH, OW,
         _dtype_code(s2d), stream_ptr())
    return y


# ---- the paired-fp16 head of DIR_FP16P (csrc/conv_pair.hip) ----------------------------------------------------------
def split_pair(t):
    """fp32 tensor -> (hi, lo) fp16 planes with hi = fp16(t), lo = fp16(t - hi): the storage format of the paired
    head (device-side torch casts: test tooling for operands, the kernels produce their own pairs)."""
    hi = t.to(torch.float16)
    return hi, (t - hi.to(torch.float32)).to(torch.float16)


def conv_bn_act_pair(x, w, bias, res=None, stride=1, pad=0, relu=True, pair_out=True):
    """dir_conv_bn_act_pair.  x = (hi, lo) or a single fp16 tensor, NHWC [B,H,W,Cin]; w = (hi, lo) [Cout,R,S,Cin];
    res = None, a single fp16 tensor or (hi, lo).  Returns (y_hi, y_lo), y_lo None when pair_out is False."""
    x_hi, x_lo = x if isinstance(x, (tuple, list)) else (x, None)
    w_hi, w_lo = w
    r_hi, r_lo = (res if isinstance(res, (tuple, list)) else (res, None))
    _need_cuda(x_hi, x_lo, w_hi, w_lo, bias, r_hi, r_lo)
    for t in (x_hi, x_lo, w_hi, w_lo, r_hi, r_lo):
        if t is not None and t.dtype != torch.float16:
            raise TypeError('fp16 planes expected')
    B, H, W, Cin = x_hi.shape
    Cout, R, S, Cin2 = w_hi.shape
    if Cin2 != Cin:
        raise ValueError('Cin mismatch')
    OH = (H + 2 * pad - R) // stride + 1
    OW = (W + 2 * pad - S) // stride + 1
    y_hi = torch.empty(B, OH, OW, Cout, dtype=torch.float16, device=x_hi.device)
    y_lo = torch.empty_like(y_hi) if pair_out else None
    call('dir_conv_bn_act_pair', ptr(x_hi), ptr(x_lo), ptr(w_hi), ptr(w_lo), ptr(bias), ptr(r_hi), ptr(r_lo),
         ptr(y_hi), ptr(y_lo), B, H, W, Cin, Cout, R, S, stride, pad, OH, OW, int(bool(relu)), stream_ptr())
    return y_hi, y_lo


def conv_pair_dual(t2, x, wcat, bias, relu=True, pair_out=True):
    """dir_conv_pair_dual: y = act([w3 | wds] . [t2 ; x] + bias); t2, x = (hi, lo) [B,H,W,Cin], wcat = (hi, lo)
    [Cout, 2 Cin] (or [Cout,1,1,2 Cin]).  Returns (y_hi, y_lo)."""
    (t_hi, t_lo), (x_hi, x_lo), (w_hi, w_lo) = t2, x, wcat
    _need_cuda(t_hi, t_lo, x_hi, x_lo, w_hi, w_lo, bias)
    B, H, W, Cin = t_hi.shape
    Cout = w_hi.shape[0]
    if w_hi.numel() != Cout * 2 * Cin or x_hi.shape != t_hi.shape:
        raise ValueError('shape mismatch')
    y_hi = torch.empty(B, H, W, Cout, dtype=torch.float16, device=t_hi.device)
    y_lo = torch.empty_like(y_hi) if pair_out else None
    call('dir_conv_pair_dual', ptr(t_hi), ptr(t_lo), ptr(x_hi), ptr(x_lo), ptr(w_hi), ptr(w_lo), ptr(bias), ptr(y_hi),
         ptr(y_lo), B, H, W, Cin, Cout, int(bool(relu)), stream_ptr())
    return y_hi, y_lo


def prep_input_pair(img, mean=None, std=None):
    """fp32 NCHW (normalised) or uint8 NHWC image batch -> space-to-depth NHWC16 stem input as an fp16 pair."""
    _need_cuda(img)
    if img.dtype == torch.float32:
        B, C, H, W = img.shape
        fmt = _lib.DIR_IMG_F32_NCHW
    elif img.dtype == torch.uint8:
        B, H, W, C = img.shape
        fmt = _lib.DIR_IMG_U8_NHWC
    else:
        raise TypeError('image must be float32 NCHW or uint8 NHWC')
    if C != 3:
        raise ValueError('3-channel image expected')
    hi = torch.empty(B, (H + 1) // 2, (W + 1) // 2, 16, dtype=torch.float16, device=img.device)
    lo = torch.empty_like(hi)
    m = (ctypes.c_float * 3)(*(mean or (0, 0, 0)))
    s = (ctypes.c_float * 3)(*(std or (1, 1, 1)))
    call('dir_prep_input_pair', ptr(img), fmt, m, s, ptr(hi), ptr(lo), B, H, W, stream_ptr())
    return hi, lo


def stem_pool_pair(s2d, w_packed, bias, out_hw):
    """dir_stem_pool_pair: s2d = (hi, lo) [B,H2,W2,16], w_packed = (hi, lo) [64,4,4,16] (pack_stem_weight of the
    fp32 filter, then split_pair), out_hw = conv output size -> pooled (hi, lo) [B,PH,PW,64]."""
    (s_hi, s_lo), (w_hi, w_lo) = s2d, w_packed
    _need_cuda(s_hi, s_lo, w_hi, w_lo, bias)
    B, H2, W2, _ = s_hi.shape
    OH, OW = out_hw
    PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    y_hi = torch.empty(B, PH, PW, 64, dtype=torch.float16, device=s_hi.device)
    y_lo = torch.empty_like(y_hi)
    call('dir_stem_pool_pair', ptr(s_hi), ptr(s_lo), ptr(w_hi), ptr(w_lo), ptr(bias), ptr(y_hi), ptr(y_lo), B, H2, W2,
         OH, OW, stream_ptr())
    return y_hi, y_lo


def stem_pool_u8(img_u8, w_oihw, bn_scale, bn_bias, mean, std, seg_tiles=0):
    """dir_stem_pool_u8: raw uint8 NHWC image [B,H,W,3] (device) + conv1.weight [64,3,7,7], the folded bn1 scale / bias (host
    fp32) and the preprocess mean / std -> pooled (hi, lo) [B,PH,PW,64]: ToTensor + Normalize + conv 7x7 s2 + BN + ReLU +
    MaxPool 3x3 s2 with the normalisation folded into the filter pair (transforms.py:617-623, resnet.py:115-119)."""
    _need_cuda(img_u8)
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 4 or img_u8.shape[3] != 3:
        raise TypeError('image must be uint8 NHWC [B,H,W,3]')
    B, H, W, _ = img_u8.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    w = w_oihw.detach().to('cpu', torch.float32).contiguous()
    sc = bn_scale.detach().to('cpu', torch.float32).contiguous()
    bi = bn_bias.detach().to('cpu', torch.float32).contiguous()
    assert tuple(w.shape) == (64, 3, 7, 7) and sc.numel() == 64 and bi.numel() == 64
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    ws = torch.empty(B * ((H + 1) // 2) * ((W + 1) // 2) * 32, dtype=torch.uint8, device=img_u8.device)
    y_hi = torch.empty(B, PH, PW, 64, dtype=torch.float16, device=img_u8.device)
    y_lo = torch.empty_like(y_hi)
    call('dir_stem_pool_u8', ptr(img_u8.contiguous()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(sc.data_ptr()),
         ctypes.c_void_p(bi.data_ptr()), m, s, ptr(ws), ptr(y_hi), ptr(y_lo), B, H, W, int(seg_tiles), stream_ptr())
    return y_hi, y_lo


def pca_whiten(X, components, mean=None, alpha=None, l2norm=False, eps_unused=None, unit_range=False):
    """out[n][j] = alpha[j] * <X[n] - mean, components[j]> (+ row L2 normalisation): dir_pca_whiten_l2, the PCA projection of
    common.whiten_features (dirtorch/utils/common.py:221-239).  unit_range=True (dir_pca_whiten_l2_unit): the caller guarantees
    |X - mean| < 64 and |components| < 64; sets of >= 32768 rows with a width that is a multiple of 32 then run as two fp16
    planes per operand on the matrix cores (csrc/sim_split.hip whiten_split_kernel) instead of the exact fp32 MFMA chain."""
    _need_cuda(X, components, mean, alpha)
    for t in (X, components, mean, alpha):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise TypeError('contiguous float32 tensors expected')
    N, D = X.shape
    v, D2 = components.shape
    if D != D2:
        raise ValueError('inner dimensions differ')
    out = torch.empty(N, v, dtype=torch.float32, device=X.device)
    if N == 0:
        return out
    call('dir_pca_whiten_l2_unit' if unit_range else 'dir_pca_whiten_l2', ptr(X), N, D, ptr(mean), ptr(components), v, ptr(alpha),
         int(bool(l2norm)), ptr(out), stream_ptr())
    return out
