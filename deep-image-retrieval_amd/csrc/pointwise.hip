// pointwise.hip — the HBM-bound kernels around the convolution trunk (gfx950).
//   prep_input        image (fp32 NCHW normalised | uint8 NHWC raw) -> 2x2 space-to-depth NHWC16
//   maxpool_3x3s2     dirtorch/nets/backbones/resnet.py:119
//   global_pool       GeM / max / avg over H*W (dirtorch/nets/layers/pooling.py:38-40,
//                     dirtorch/nets/rmac_resnet.py:24-31,52-59)
//   upsample_add      x4 + nearest-upsampled lateral map (dirtorch/nets/rmac_resnet_fpn.py:55-60)
//   l2norm_rows       F.normalize(p=2) (dirtorch/nets/rmac_resnet.py:7-9)
//   multiscale_pool   dirtorch/utils/common.py:41-55
// All of them move 16 bytes per lane and keep every reduction in fp32.
#include "dir_common.h"
#include "pointwise.h"

namespace dir {

// ---- prep_input -------------------------------------------------------------------------------
// out[b][y2][x2][(dy*2+dx)*3 + c] = pixel (2*y2+dy, 2*x2+dx) channel c; channels 12..15 = 0.
template <class DT, int FMT>
__global__ void prep_input_kernel(const void* __restrict__ img, uint16_t* __restrict__ out, int B,
                                  int H, int W, int H2, int W2, float m0, float m1, float m2,
                                  float s0, float s1, float s2) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * H2 * W2;
    if (idx >= total) return;
    const int x2 = (int)(idx % W2);
    const int y2 = (int)((idx / W2) % H2);
    const int b = (int)(idx / ((long)W2 * H2));
    const float mean[3] = {m0, m1, m2};
    const float stdv[3] = {s0, s1, s2};
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int y = 2 * y2 + dy;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int x = 2 * x2 + dx;
            if (y < H && x < W) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float f;
                    if (FMT == DIR_IMG_F32_NCHW) {
                        f = ((const float*)img)[(((size_t)b * 3 + c) * H + y) * W + x];
                    } else {
                        // ToTensor (/255) then Normalize ((x-mean)/std):
                        // dirtorch/utils/transforms.py:617-623
                        const uint8_t u = ((const uint8_t*)img)[(((size_t)b * H + y) * W + x) * 3 + c];
                        f = ((float)u / 255.f - mean[c]) / stdv[c];
                    }
                    v[(dy * 2 + dx) * 3 + c] = f;
                }
            }
        }
    }
    u32x4_t o0, o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o0[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
        o1[e] = pack2<DT>(v[8 + 2 * e], v[8 + 2 * e + 1]);
    }
    gstore16(out + idx * 16, o0);
    gstore16(out + idx * 16 + 8, o1);
}

int prep_input(const void* img, int fmt, const float* mean3, const float* std3, void* out, int B,
               int H, int W, int dtype, hipStream_t stream) {
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const long total = (long)B * H2 * W2;
    const int threads = 256;
    const unsigned blocks = (unsigned)((total + threads - 1) / threads);
    float m[3] = {0, 0, 0}, s[3] = {1, 1, 1};
    if (fmt == DIR_IMG_U8_NHWC) {
        if (!mean3 || !std3) return fail(DIR_ERR_INVALID, "prep_input: u8 input needs mean/std");
        for (int c = 0; c < 3; ++c) {
            m[c] = mean3[c];
            s[c] = std3[c];
        }
    } else if (fmt != DIR_IMG_F32_NCHW) {
        return fail(DIR_ERR_INVALID, "prep_input: bad image format");
    }
#define DIR_PREP(DT, FMT)                                                                         \
    hipLaunchKernelGGL((prep_input_kernel<DT, FMT>), dim3(blocks), dim3(threads), 0, stream, img, \
                       (uint16_t*)out, B, H, W, H2, W2, m[0], m[1], m[2], s[0], s[1], s[2])
    if (dtype == DIR_BF16) {
        if (fmt == DIR_IMG_F32_NCHW) DIR_PREP(BF16, DIR_IMG_F32_NCHW);
        else DIR_PREP(BF16, DIR_IMG_U8_NHWC);
    } else if (dtype == DIR_FP16) {
        if (fmt == DIR_IMG_F32_NCHW) DIR_PREP(FP16, DIR_IMG_F32_NCHW);
        else DIR_PREP(FP16, DIR_IMG_U8_NHWC);
    } else {
        return fail(DIR_ERR_INVALID, "prep_input: bad dtype");
    }
#undef DIR_PREP
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

// ---- maxpool 3x3 stride 2 pad 1 ------------------------------------------------------------------
template <class DT>
__global__ void maxpool_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int B,
                               int H, int W, int C, int PH, int PW) {
    const int C8 = C >> 3;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * PH * PW * C8;
    if (idx >= total) return;
    const int c8 = (int)(idx % C8);
    const int pw = (int)((idx / C8) % PW);
    const int ph = (int)((idx / ((long)C8 * PW)) % PH);
    const int b = (int)(idx / ((long)C8 * PW * PH));
    float best[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int ih = 2 * ph - 1 + r;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int iw = 2 * pw - 1 + s;
            if ((unsigned)iw >= (unsigned)W) continue;
            const u32x4_t v = gload16(x + (((size_t)b * H + ih) * W + iw) * C + c8 * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float lo, hi;
                unpack2<DT>(v[e], lo, hi);
                best[2 * e] = fmaxf(best[2 * e], lo);
                best[2 * e + 1] = fmaxf(best[2 * e + 1], hi);
            }
        }
    }
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(best[2 * e], best[2 * e + 1]);
    gstore16(y + idx * 8, o);
}

int maxpool_3x3s2(const void* x, void* y, int B, int H, int W, int C, int dtype,
                  hipStream_t stream) {
    if (C % 8 != 0) return fail(DIR_ERR_INVALID, "maxpool: C must be a multiple of 8");
    const int PH = (H - 1) / 2 + 1, PW = (W - 1) / 2 + 1;  // floor((H + 2 - 3) / 2) + 1
    const long total = (long)B * PH * PW * (C / 8);
    const int threads = 256;
    const unsigned blocks = (unsigned)((total + threads - 1) / threads);
    if (dtype == DIR_BF16)
        hipLaunchKernelGGL(maxpool_kernel<BF16>, dim3(blocks), dim3(threads), 0, stream,
                           (const uint16_t*)x, (uint16_t*)y, B, H, W, C, PH, PW);
    else if (dtype == DIR_FP16)
        hipLaunchKernelGGL(maxpool_kernel<FP16>, dim3(blocks), dim3(threads), 0, stream,
                           (const uint16_t*)x, (uint16_t*)y, B, H, W, C, PH, PW);
    else
        return fail(DIR_ERR_INVALID, "maxpool: bad dtype");
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

// ---- global pooling (GeM / max / avg) -------------------------------------------------------------
// One workgroup = one image x 64 channels: 8 lanes span the 64 channels (16 B each, one 128-byte
// line per pixel), 32 lane-rows stride over the pixels; fp32 partials meet in LDS.
__device__ inline float center_mask(int h, int w, int H, int W, float cb) {
    // 1 + bilinear(align_corners=True) upsampling of the 4x4 mask with cb in the middle 2x2
    // (dirtorch/nets/rmac_resnet.py:52-56).
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

template <class DT, int POOL>
__global__ void __launch_bounds__(256) global_pool_kernel(const uint16_t* __restrict__ x,
                                                         float* __restrict__ out, int ldo, int H,
                                                         int W, int C, float p, float eps, float cb) {
    __shared__ float part[32][64 + 1];
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    const int cl = threadIdx.x & 7;   // which 8-channel group
    const int pr = threadIdx.x >> 3;  // pixel lane-row 0..31
    const int HW = H * W;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = (POOL == DIR_POOL_MAX) ? -INFINITY : 0.f;
    const uint16_t* base = x + (size_t)b * HW * C + c0 + cl * 8;
    // x^p for x >= eps > 0: p == 3 (the usual GeM exponent) is two multiplies, anything else
    // exp2(p * log2 x) on the transcendental unit (inputs are clamped, so no denormal handling)
    const bool cube = (p == 3.f);
    auto powp = [&](float t) {
        return cube ? t * t * t : __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(t));
    };
    constexpr int UNR = 4;  // 4 independent 16-byte loads in flight per lane
    for (int px0 = pr; px0 < HW; px0 += 32 * UNR) {
        u32x4_t v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int px = px0 + 32 * u;
            v[u] = gload16(base + (size_t)(px < HW ? px : pr) * C);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int px = px0 + 32 * u;
            if (px >= HW) break;
            const float mk = cb > 0.f ? center_mask(px / W, px % W, H, W, cb) : 1.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float f[2];
                DT::unpack(v[u][e], f[0], f[1]);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float t = f[q] * mk;
                    if (POOL == DIR_POOL_GEM)
                        acc[2 * e + q] += powp(fmaxf(t, eps));
                    else if (POOL == DIR_POOL_MAX)
                        acc[2 * e + q] = fmaxf(acc[2 * e + q], t);
                    else
                        acc[2 * e + q] += t;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[pr][cl * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float r = (POOL == DIR_POOL_MAX) ? -INFINITY : 0.f;
        for (int i = 0; i < 32; ++i)
            r = (POOL == DIR_POOL_MAX) ? fmaxf(r, part[i][threadIdx.x]) : r + part[i][threadIdx.x];
        if (POOL == DIR_POOL_GEM)
            r = powf(r / (float)HW, 1.f / p);
        else if (POOL == DIR_POOL_AVG)
            r = r / (float)HW;
        out[(size_t)b * ldo + c0 + threadIdx.x] = r;
    }
}

int global_pool(const void* x, float* out, int ldo, int B, int H, int W, int C, int pooling, float p,
                float eps, float center_bias, int dtype, hipStream_t stream) {
    if (C % 64 != 0) return fail(DIR_ERR_INVALID, "global_pool: C must be a multiple of 64");
    if (pooling == DIR_POOL_GEM && !(p > 0.f)) return fail(DIR_ERR_INVALID, "global_pool: p <= 0");
    if (dtype != DIR_BF16 && dtype != DIR_FP16) return fail(DIR_ERR_INVALID, "global_pool: bad dtype");
    const dim3 grid(C / 64, B);
#define DIR_GP(DT, POOL)                                                                      \
    hipLaunchKernelGGL((global_pool_kernel<DT, POOL>), grid, dim3(256), 0, stream,            \
                       (const uint16_t*)x, out, ldo, H, W, C, p, eps, center_bias)
    if (pooling == DIR_POOL_GEM) {
        if (dtype == DIR_BF16) DIR_GP(BF16, DIR_POOL_GEM); else DIR_GP(FP16, DIR_POOL_GEM);
    } else if (pooling == DIR_POOL_MAX) {
        if (dtype == DIR_BF16) DIR_GP(BF16, DIR_POOL_MAX); else DIR_GP(FP16, DIR_POOL_MAX);
    } else if (pooling == DIR_POOL_AVG) {
        if (dtype == DIR_BF16) DIR_GP(BF16, DIR_POOL_AVG); else DIR_GP(FP16, DIR_POOL_AVG);
    } else {
        return fail(DIR_ERR_INVALID, "global_pool: bad pooling mode");
    }
#undef DIR_GP
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

// ---- upsample_add ------------------------------------------------------------------------------
// y[b][Y][X][:] = x[b][Y][X][:] + low[b][sy(Y)][sx(X)][:], the x4 + F.interpolate(c5, size=x4.shape[-2:],
// mode='nearest') of dirtorch/nets/rmac_resnet_fpn.py:55-60.  Source index as PyTorch computes it:
// min(floor(dst * (float)in / out), in - 1).
template <class DT>
__global__ void upsample_add_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ low,
                                    uint16_t* __restrict__ y, long total, int H, int W, int h, int w,
                                    int C8, float sy, float sx, int* ovf_flag) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx % C8);
    const long pix = idx / C8;
    const int X = (int)(pix % W);
    const int Y = (int)((pix / W) % H);
    const long b = pix / ((long)W * H);
    const int ys = min((int)floorf(Y * sy), h - 1);
    const int xs = min((int)floorf(X * sx), w - 1);
    const u32x4_t a = gload16(x + idx * 8);
    const u32x4_t l = gload16(low + (((b * h + ys) * w + xs) * C8 + c8) * 8);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a0, a1, l0, l1;
        DT::unpack(a[e], a0, a1);
        DT::unpack(l[e], l0, l1);
        o[e] = DT::pack(a0 + l0, a1 + l1);
    }
    gstore16(y + idx * 8, o);
    Ovf<DT> ovf;
    ovf.see(o);
    ovf.flush(ovf_flag);
}

int upsample_add(const void* x, const void* low, void* y, int B, int H, int W, int h, int w, int C,
                 int dtype, hipStream_t stream, int* ovf) {
    if (C % 8 != 0) return fail(DIR_ERR_INVALID, "upsample_add: C must be a multiple of 8");
    if (dtype != DIR_BF16 && dtype != DIR_FP16) return fail(DIR_ERR_INVALID, "upsample_add: bad dtype");
    const long total = (long)B * H * W * (C / 8);
    const dim3 grid((unsigned)((total + 255) / 256));
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    if (dtype == DIR_BF16)
        hipLaunchKernelGGL(upsample_add_kernel<BF16>, grid, dim3(256), 0, stream, (const uint16_t*)x,
                           (const uint16_t*)low, (uint16_t*)y, total, H, W, h, w, C / 8, sy, sx, ovf);
    else
        hipLaunchKernelGGL(upsample_add_kernel<FP16>, grid, dim3(256), 0, stream, (const uint16_t*)x,
                           (const uint16_t*)low, (uint16_t*)y, total, H, W, h, w, C / 8, sy, sx, ovf);
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

// ---- row L2 normalisation ---------------------------------------------------------------------
__global__ void __launch_bounds__(256) l2norm_rows_kernel(float* __restrict__ x, int cols,
                                                         float eps) {
    __shared__ float wsum[4];
    float* row = x + (size_t)blockIdx.x * cols;
    float ss = 0.f;
    for (int i = threadIdx.x; i < cols; i += 256) {
        const float v = row[i];
        ss += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const float inv = 1.f / fmaxf(sqrtf(tot), eps);
    for (int i = threadIdx.x; i < cols; i += 256) row[i] *= inv;
}

int l2norm_rows(float* x, int rows, int cols, float eps, hipStream_t stream) {
    if (rows <= 0) return DIR_OK;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(rows), dim3(256), 0, stream, x, cols, eps);
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

// ---- multi-scale descriptor pooling ---------------------------------------------------------------
__global__ void multiscale_pool_kernel(const float* __restrict__ x, float* __restrict__ out, int S,
                                       long ND, int mode, float gemp) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ND) return;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) {
        const float v = x[(size_t)s * ND + idx];
        if (mode == 0) {
            acc += v;
        } else {
            // sympow(x, p) = sign(x) * max(|x|, 1e-6)^p with sign(0) = 0 (common.py:48-50)
            const float sg = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
            acc += powf(fmaxf(v * sg, 1e-6f), gemp) * sg;
        }
    }
    acc /= (float)S;
    if (mode != 0) {
        const float sg = acc > 0.f ? 1.f : (acc < 0.f ? -1.f : 0.f);
        acc = powf(fmaxf(acc * sg, 1e-6f), 1.f / gemp) * sg;
    }
    out[idx] = acc;
}

int multiscale_pool(const float* x, float* out, int S, int N, int D, int mode, float gemp,
                    hipStream_t stream) {
    if (S <= 0) return fail(DIR_ERR_INVALID, "multiscale_pool: S <= 0");
    if (mode != 0 && mode != 1) return fail(DIR_ERR_INVALID, "multiscale_pool: bad mode");
    const long ND = (long)N * D;
    if (ND == 0) return DIR_OK;
    const unsigned blocks = (unsigned)((ND + 255) / 256);
    hipLaunchKernelGGL(multiscale_pool_kernel, dim3(blocks), dim3(256), 0, stream, x, out, S, ND,
                       mode, gemp);
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

}  // namespace dir
