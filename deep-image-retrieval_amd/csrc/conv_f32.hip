// conv_f32.hip — the STRICT path: the trunk in the reference's own arithmetic (fp32 storage, fp32 products,
// fp32 accumulation) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// Why it exists.  The north-star tolerance is 1e-4 cosine against the reference's fp32 CPU path
// (dirtorch/nets/backbones/resnet.py:67-87,157-174).  16-bit storage cannot promise that on a network whose
// BatchNorm layers cancel a large common mode: on the BatchNorm-calibrated test checkpoint an IDEAL fp16
// implementation loses 0.9e-4 (ResNet-101 @ 1024^2) / 1.3e-4 (ResNet-50 @ 224^2): rounding the folded WEIGHTS alone
// costs 6.8e-5 of the latter, the residual carry 5.0e-5, the branch activations 1.8e-5 - an fp32 residual carry would
// only bring fp16 to 8.5e-5 and does nothing for bf16 (tests/precision_decomposition.py).  This
// path removes all three: every conv is an implicit GEMM over fp32 NHWC
// activations and fp32 [Cout][R][S][Cin] weights (eval-mode BatchNorm folded in fp32), bias / residual / ReLU in
// the epilogue, nothing fused across layers.  The f32 MFMA is a k-ordered fmaf chain, so a conv output differs
// from the reference's only by summation order (~1e-7 relative).
//
// Cost: the fp32 matrix pipe peaks at 157 TFLOP/s (1/16 of bf16) - the whole path is MFMA-bound, which is why the
// kernel is deliberately plain: 128 pixels x 128 (or 64) channels per workgroup, K slabs of 32 floats staged
// through registers into swizzled LDS (the tile loop of gemm_f32.hip with an im2col gather in front of it), 3
// workgroups per CU hide the fetch latency.  Selected with dir_engine_finalize(DIR_F32) /
// DIRTORCH_AMD_DTYPE=f32; bench.py reports its rate next to the 16-bit ones.
//
// Also here: the fp32 forms of the HBM-bound kernels around the trunk (space-to-depth input, max pool, global
// pooling, FPN upsample-add).
#include "conv_f32.h"

namespace dir {

template <int BN>
__global__ void __launch_bounds__(256) conv_f32_kernel(const ConvF32Args a) {
    constexpr int BM = 128;
    constexpr int TN = 2;                       // 32-channel tiles per wave
    constexpr int TM = BN == 128 ? 2 : 1;       // 32-pixel tiles per wave
    constexpr int WROWS = BN / 32;              // 32-row groups of the weight slab per thread
    constexpr int XS = BM * 128;                // bytes of the pixel slab (128 rows x 32 floats)
    __shared__ __attribute__((aligned(16))) char smem[(BM + BN) * 128];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n;          // n fastest: neighbours share the pixel slab in L2
    const int tile_m = wg / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int slot = tid & 7;
    const int srcchunk = slot ^ ((tid >> 4) & 7);   // 16-byte chunk (4 floats) of the 128-byte LDS row
    const int prow = tid >> 3;                      // + 32 * i

    // ---- per-row constants of the im2col gather (4 pixel rows per thread) -------------------------------
    int ih0[4], iw0[4];
    long pix0[4];                                   // b * H * W, or -1 for rows past M
    const int OHW = a.OH * a.OW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + i * 32 + prow;
        if (m < a.M) {
            const int b = m / OHW, rem = m - b * OHW, oh = rem / a.OW, ow = rem - oh * a.OW;
            ih0[i] = oh * a.stride - a.pad;
            iw0[i] = ow * a.stride - a.pad;
            pix0[i] = (long)b * a.H * a.W;
        } else {
            ih0[i] = iw0[i] = 0;
            pix0[i] = -1;
        }
    }

    f32x4_t xr[4], wr[WROWS];
    auto fetch = [&](int k0) {
        const int k = k0 + srcchunk * 4;
        const bool kok = k < a.Ktot;
        const int tap = k / a.Cin, c = k - tap * a.Cin;     // a 4-float chunk never straddles a tap (Cin % 4 == 0)
        const int r = tap / a.S, s = tap - r * a.S;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ih = ih0[i] + r, iw = iw0[i] + s;
            const bool ok = kok && pix0[i] >= 0 && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            f32x4_t v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(const DIR_GLOBAL f32x4_t*)(a.x + (size_t)(pix0[i] + (long)ih * a.W + iw) * a.Cin + c);
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            const int n = n0 + i * 32 + prow;
            f32x4_t v = {0.f, 0.f, 0.f, 0.f};
            if (kok && n < a.Cout) v = *(const DIR_GLOBAL f32x4_t*)(a.w + (size_t)n * a.Ktot + k);
            wr[i] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4_t*)(smem + (i * 256 + tid) * 16) = xr[i];
#pragma unroll
        for (int i = 0; i < WROWS; ++i) *(f32x4_t*)(smem + XS + (i * 256 + tid) * 16) = wr[i];
    };

    const int lrow = lane & 31, lhi = lane >> 5, lswz = (lane >> 1) & 7;
    int loff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) loff[c] = lrow * 128 + (((2 * c + lhi) ^ lswz) << 4);
    const int wm = BN == 128 ? (wave & 1) : wave;           // wave grid: 2 x 2 (BN 128) or 4 x 1 (BN 64)
    const int wn = BN == 128 ? (wave >> 1) : 0;
    const int xbase = wm * (TM * 32) * 128;
    const int wbase = XS + wn * (TN * 32) * 128;

    f32x16_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int T = (a.Ktot + 31) / 32;
    fetch(0);
    for (int t = 0; t < T; ++t) {
        __syncthreads();            // everyone is done reading the previous slab
        commit();
        __syncthreads();
        if (t + 1 < T) fetch((t + 1) * 32);     // in flight under this slab's MFMAs
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4_t wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = *(const f32x4_t*)(smem + wbase + i * 4096 + loff[c]);
#pragma unroll
            for (int j = 0; j < TM; ++j) xf[j] = *(const f32x4_t*)(smem + xbase + j * 4096 + loff[c]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[i][q], xf[j][q], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: A = weights, so a lane owns 4 consecutive channels (8g + 4 lhi + e) of pixel lrow -----------
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * (TM * 32) + j * 32 + lrow;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * (TN * 32) + i * 32 + 8 * g + 4 * lhi;
                if (n >= a.Cout) continue;
                const f32x4_t bz = *(const DIR_GLOBAL f32x4_t*)(a.bias + n);
                f32x4_t v = {acc[i][j][4 * g + 0] + bz[0], acc[i][j][4 * g + 1] + bz[1],
                             acc[i][j][4 * g + 2] + bz[2], acc[i][j][4 * g + 3] + bz[3]};
                const size_t o = (size_t)m * a.Cout + n;
                if (a.res) v = v + *(const DIR_GLOBAL f32x4_t*)(a.res + o);
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];   // torch's relu: a NaN stays a NaN (fmaxf would flush it)
                }
                *(DIR_GLOBAL f32x4_t*)(a.y + o) = v;
            }
    }
}

int conv_f32_launch(ConvF32Args a, hipStream_t stream, const char** variant) {
    if (a.Cin % 4 != 0 || a.Cout % 4 != 0)
        return fail(DIR_ERR_INVALID, "conv_f32: Cin and Cout must be multiples of 4");
    if (a.B <= 0 || a.M != (long)a.B * a.OH * a.OW || a.Ktot != a.R * a.S * a.Cin)
        return fail(DIR_ERR_INVALID, "conv_f32: inconsistent shape");
    if ((long)a.B * a.OH * a.OW >= (1L << 31)) return fail(DIR_ERR_INVALID, "conv_f32: too many output pixels");
    const bool narrow = a.Cout <= 64;
    const int BN = narrow ? 64 : 128;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    const long tiles = (long)((a.M + 127) / 128) * a.tiles_n;
    if (tiles >= (1L << 31)) return fail(DIR_ERR_INVALID, "conv_f32: too many tiles");
    if (variant) *variant = narrow ? "128x64" : "128x128";
    if (narrow)
        hipLaunchKernelGGL(conv_f32_kernel<64>, dim3((unsigned)tiles), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(conv_f32_kernel<128>, dim3((unsigned)tiles), dim3(256), 0, stream, a);
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

// ---- prep_input: image -> 2x2 space-to-depth NHWC fp32, 12 real + 4 zero channels ---------------------------
// Same layout as the 16-bit path (pointwise.hip prep_input_kernel): the 7x7 stride-2 stem becomes a 4x4 stride-1
// conv whose K slab of 32 floats is two whole taps.  ToTensor / Normalize of a uint8 feed are fused here
// (dirtorch/utils/transforms.py:617-623).
template <int FMT>
__global__ void prep_input_f32_kernel(const void* __restrict__ img, float* __restrict__ out, int B, int H, int W,
                                      int H2, int W2, float m0, float m1, float m2, float s0, float s1, float s2) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * H2 * W2 * 4;       // one (dy, dx) sub-pixel = one float4 per thread
    if (idx >= total) return;
    const int sub = (int)(idx & 3);
    const long p = idx >> 2;
    const int x2 = (int)(p % W2);
    const int y2 = (int)((p / W2) % H2);
    const int b = (int)(p / ((long)W2 * H2));
    const int y = 2 * y2 + (sub >> 1), x = 2 * x2 + (sub & 1);
    const float mean[3] = {m0, m1, m2};
    const float stdv[3] = {s0, s1, s2};
    float v[3] = {0.f, 0.f, 0.f};
    if (y < H && x < W) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (FMT == DIR_IMG_F32_NCHW) {
                v[c] = ((const float*)img)[(((size_t)b * 3 + c) * H + y) * W + x];
            } else {
                const uint8_t u = ((const uint8_t*)img)[(((size_t)b * H + y) * W + x) * 3 + c];
                v[c] = ((float)u / 255.f - mean[c]) / stdv[c];
            }
        }
    }
    // channel (dy*2+dx)*3 + c of the 16-float pixel; channels 12..15 are written (as zeros) by sub-pixel 3's
    // neighbour store below
    float* o = out + p * 16 + sub * 3;
    o[0] = v[0];
    o[1] = v[1];
    o[2] = v[2];
    if (sub == 3) {
        o[3] = 0.f;
        o[4] = 0.f;
        o[5] = 0.f;
        o[6] = 0.f;
    }
}

int prep_input_f32(const void* img, int fmt, const float* mean3, const float* std3, float* out, int B, int H, int W,
                   hipStream_t stream) {
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const long total = (long)B * H2 * W2 * 4;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    float m[3] = {0, 0, 0}, s[3] = {1, 1, 1};
    if (fmt == DIR_IMG_U8_NHWC) {
        if (!mean3 || !std3) return fail(DIR_ERR_INVALID, "prep_input: u8 input needs mean/std");
        for (int c = 0; c < 3; ++c) {
            m[c] = mean3[c];
            s[c] = std3[c];
        }
        hipLaunchKernelGGL(prep_input_f32_kernel<DIR_IMG_U8_NHWC>, dim3(blocks), dim3(256), 0, stream, img, out, B, H,
                           W, H2, W2, m[0], m[1], m[2], s[0], s[1], s[2]);
    } else if (fmt == DIR_IMG_F32_NCHW) {
        hipLaunchKernelGGL(prep_input_f32_kernel<DIR_IMG_F32_NCHW>, dim3(blocks), dim3(256), 0, stream, img, out, B,
                           H, W, H2, W2, m[0], m[1], m[2], s[0], s[1], s[2]);
    } else {
        return fail(DIR_ERR_INVALID, "prep_input: bad image format");
    }
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

// ---- maxpool 3x3 stride 2 pad 1 (dirtorch/nets/backbones/resnet.py:119), 4 channels per lane ------------------
__global__ void maxpool_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C,
                                   int PH, int PW) {
    const int C4 = C >> 2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * PH * PW * C4;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    const int pw = (int)((idx / C4) % PW);
    const int ph = (int)((idx / ((long)C4 * PW)) % PH);
    const int b = (int)(idx / ((long)C4 * PW * PH));
    f32x4_t best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int ih = 2 * ph - 1 + r;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int iw = 2 * pw - 1 + s;
            if ((unsigned)iw >= (unsigned)W) continue;
            const f32x4_t v = *(const DIR_GLOBAL f32x4_t*)(x + (((size_t)b * H + ih) * W + iw) * C + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) best[e] = fmaxf(best[e], v[e]);
        }
    }
    *(DIR_GLOBAL f32x4_t*)(y + idx * 4) = best;
}

int maxpool_3x3s2_f32(const float* x, float* y, int B, int H, int W, int C, hipStream_t stream) {
    if (C % 4 != 0) return fail(DIR_ERR_INVALID, "maxpool: C must be a multiple of 4");
    const int PH = (H - 1) / 2 + 1, PW = (W - 1) / 2 + 1;
    const long total = (long)B * PH * PW * (C / 4);
    hipLaunchKernelGGL(maxpool_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, y, B, H, W,
                       C, PH, PW);
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

// ---- global pooling (GeM / max / avg, dirtorch/nets/layers/pooling.py:38-40, rmac_resnet.py:52-59) ---------------
// One workgroup = one image x 64 channels: 16 lanes span the channels (float4 each), 16 lane-rows stride over the
// pixels; partials meet in LDS.  Same arithmetic as the 16-bit kernel (pointwise.hip), fp32 in.
__device__ inline float center_mask_f32(int h, int w, int H, int W, float cb) {
    const float sy = H > 1 ? (float)h * 3.f / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)w * 3.f / (float)(W - 1) : 0.f;
    int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
    y0 = y0 > 2 ? 2 : y0;
    x0 = x0 > 2 ? 2 : x0;
    const float fy = sy - (float)y0, fx = sx - (float)x0;
    auto mval = [&](int y, int x) { return (y >= 1 && y <= 2 && x >= 1 && x <= 2) ? cb : 0.f; };
    const float top = mval(y0, x0) * (1.f - fx) + mval(y0, x0 + 1) * fx;
    const float bot = mval(y0 + 1, x0) * (1.f - fx) + mval(y0 + 1, x0 + 1) * fx;
    return 1.f + top * (1.f - fy) + bot * fy;
}

template <int POOL>
__global__ void __launch_bounds__(256) global_pool_f32_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                             int ldo, int H, int W, int C, float p, float eps,
                                                             float cb) {
    __shared__ float part[16][64 + 1];
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    const int cl = threadIdx.x & 15;  // which 4-channel group
    const int pr = threadIdx.x >> 4;  // pixel lane-row 0..15
    const int HW = H * W;
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = (POOL == DIR_POOL_MAX) ? -INFINITY : 0.f;
    const float* base = x + (size_t)b * HW * C + c0 + cl * 4;
    for (int px = pr; px < HW; px += 16) {
        const f32x4_t v = *(const DIR_GLOBAL f32x4_t*)(base + (size_t)px * C);
        const float mk = cb > 0.f ? center_mask_f32(px / W, px % W, H, W, cb) : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = v[e] * mk;
            if (POOL == DIR_POOL_GEM)
                acc[e] += powf(fmaxf(t, eps), p);       // clamp(min=eps).pow(p), pooling.py:39
            else if (POOL == DIR_POOL_MAX)
                acc[e] = fmaxf(acc[e], t);
            else
                acc[e] += t;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[pr][cl * 4 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float r = (POOL == DIR_POOL_MAX) ? -INFINITY : 0.f;
        for (int i = 0; i < 16; ++i)
            r = (POOL == DIR_POOL_MAX) ? fmaxf(r, part[i][threadIdx.x]) : r + part[i][threadIdx.x];
        if (POOL == DIR_POOL_GEM)
            r = powf(r / (float)HW, 1.f / p);
        else if (POOL == DIR_POOL_AVG)
            r = r / (float)HW;
        out[(size_t)b * ldo + c0 + threadIdx.x] = r;
    }
}

int global_pool_f32(const float* x, float* out, int ldo, int B, int H, int W, int C, int pooling, float p, float eps,
                    float center_bias, hipStream_t stream) {
    if (C % 64 != 0) return fail(DIR_ERR_INVALID, "global_pool: C must be a multiple of 64");
    if (pooling == DIR_POOL_GEM && !(p > 0.f)) return fail(DIR_ERR_INVALID, "global_pool: p <= 0");
    const dim3 grid(C / 64, B);
    if (pooling == DIR_POOL_GEM)
        hipLaunchKernelGGL(global_pool_f32_kernel<DIR_POOL_GEM>, grid, dim3(256), 0, stream, x, out, ldo, H, W, C, p,
                           eps, center_bias);
    else if (pooling == DIR_POOL_MAX)
        hipLaunchKernelGGL(global_pool_f32_kernel<DIR_POOL_MAX>, grid, dim3(256), 0, stream, x, out, ldo, H, W, C, p,
                           eps, center_bias);
    else if (pooling == DIR_POOL_AVG)
        hipLaunchKernelGGL(global_pool_f32_kernel<DIR_POOL_AVG>, grid, dim3(256), 0, stream, x, out, ldo, H, W, C, p,
                           eps, center_bias);
    else
        return fail(DIR_ERR_INVALID, "global_pool: bad pooling mode");
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

// ---- x4 + nearest-upsampled lateral map (dirtorch/nets/rmac_resnet_fpn.py:55-60) --------------------------------
__global__ void upsample_add_f32_kernel(const float* __restrict__ x, const float* __restrict__ low,
                                        float* __restrict__ y, long total, int H, int W, int h, int w, int C4,
                                        float sy, float sx) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    const long pix = idx / C4;
    const int X = (int)(pix % W);
    const int Y = (int)((pix / W) % H);
    const long b = pix / ((long)W * H);
    const int ys = min((int)floorf(Y * sy), h - 1);
    const int xs = min((int)floorf(X * sx), w - 1);
    const f32x4_t a = *(const DIR_GLOBAL f32x4_t*)(x + idx * 4);
    const f32x4_t l = *(const DIR_GLOBAL f32x4_t*)(low + (((b * h + ys) * w + xs) * C4 + c4) * 4);
    *(DIR_GLOBAL f32x4_t*)(y + idx * 4) = a + l;
}

int upsample_add_f32(const float* x, const float* low, float* y, int B, int H, int W, int h, int w, int C,
                     hipStream_t stream) {
    if (C % 4 != 0) return fail(DIR_ERR_INVALID, "upsample_add: C must be a multiple of 4");
    const long total = (long)B * H * W * (C / 4);
    hipLaunchKernelGGL(upsample_add_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, low, y,
                       total, H, W, h, w, C / 4, (float)h / (float)H, (float)w / (float)W);
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

}  // namespace dir
