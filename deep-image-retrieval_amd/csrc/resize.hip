// resize.hip — the test-time `Scale` transform on the GPU.
//
// dirtorch/utils/transforms.py:133-185 resizes with PIL: img.resize((ow, oh), Image.BILINEAR).  For an
// 8-bit RGB image Pillow (src/libImaging/Resample.c) does
//   1. per output coordinate: a triangle filter whose support is scaled by max(in/out, 1)
//      (antialiased when shrinking), weights normalised in double and rounded to 22-bit fixed point;
//   2. a horizontal pass to a uint8 image, then a vertical pass, each (1<<21 + sum w*p) >> 22, clipped;
//   3. a pass whose size does not change is skipped.
// The kernels below reproduce that arithmetic bit for bit (the tests compare against Pillow itself):
// the coefficient tables are built on the device in fp64 with contraction disabled, so one C call is
// three asynchronous launches on the caller's stream and no host table has to outlive the call.
#include "dir_common.h"
#include "pointwise.h"

namespace dir {

constexpr int kPrecisionBits = 32 - 8 - 2;

__host__ __device__ inline int resample_ksize(int in_size, int out_size) {
    double fs = (double)in_size / (double)out_size;
    if (fs < 1.0) fs = 1.0;
    return (int)ceil(1.0 * fs) * 2 + 1;
}

// tab layout per axis: int xmin[out], int count[out], int kk[out][ksize]
__global__ void resample_coeffs_kernel(int* __restrict__ tab, int in_size, int out_size, int ksize) {
#pragma clang fp contract(off)
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= out_size) return;
    const double scale = (double)in_size / (double)out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int* kk = tab + 2 * (size_t)out_size + (size_t)xx * ksize;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double a = (x + xmin - center + 0.5) * ss;
        if (a < 0.0) a = -a;
        ww += a < 1.0 ? 1.0 - a : 0.0;
    }
    for (int x = 0; x < ksize; ++x) {
        double w = 0.0;
        if (x < xmax) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            w = a < 1.0 ? 1.0 - a : 0.0;
            if (ww != 0.0) w /= ww;
        }
        kk[x] = w < 0.0 ? (int)(-0.5 + w * (double)(1 << kPrecisionBits))
                        : (int)(0.5 + w * (double)(1 << kPrecisionBits));
    }
    tab[xx] = xmin;
    tab[out_size + xx] = xmax;
}

// One pass along the axis whose element stride is `astride` bytes; the other two index levels are
// flattened by the caller: element (outer, o, inner) with `inner` bytes contiguous.
//   horizontal: outer = b*H + y, inner = 3 (channels),      astride = 3
//   vertical:   outer = b,       inner = OW*3 (a whole row), astride = OW*3
__global__ void resample_pass_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                     const int* __restrict__ tab, long total, int in_size, int out_size,
                                     int ksize, long inner) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long i = idx % inner;
    const int o = (int)((idx / inner) % out_size);
    const long outer = idx / (inner * out_size);
    const int xmin = tab[o], cnt = tab[out_size + o];
    const int* kk = tab + 2 * (size_t)out_size + (size_t)o * ksize;
    const uint8_t* p = src + (outer * in_size + xmin) * inner + i;
    int acc = 1 << (kPrecisionBits - 1);
    for (int x = 0; x < cnt; ++x) acc += (int)p[(long)x * inner] * kk[x];
    acc >>= kPrecisionBits;
    dst[idx] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
}

static inline size_t tab_ints(int in_size, int out_size) {
    return (size_t)out_size * (2 + resample_ksize(in_size, out_size));
}

size_t resize_workspace_bytes(int B, int H, int W, int OH, int OW) {
    size_t b = (tab_ints(W, OW) + tab_ints(H, OH)) * sizeof(int);
    b = (b + 255) / 256 * 256;
    return b + (size_t)B * H * OW * 3;  // the horizontally resampled intermediate image
}

int resize_bilinear_u8(const uint8_t* src, uint8_t* dst, int B, int H, int W, int OH, int OW, void* ws,
                       size_t ws_bytes, hipStream_t stream) {
    if (ws_bytes < resize_workspace_bytes(B, H, W, OH, OW) || !ws)
        return fail(DIR_ERR_WORKSPACE, "resize: workspace too small");
    if (((uintptr_t)ws & 3) != 0) return fail(DIR_ERR_INVALID, "resize: workspace must be 4-byte aligned");
    const bool horiz = OW != W, vert = OH != H;
    if (!horiz && !vert) {
        DIR_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)B * H * W * 3, hipMemcpyDeviceToDevice, stream));
        return DIR_OK;
    }
    int* xtab = (int*)ws;
    int* ytab = xtab + tab_ints(W, OW);
    uint8_t* tmp = (uint8_t*)ws + ((tab_ints(W, OW) + tab_ints(H, OH)) * sizeof(int) + 255) / 256 * 256;
    const int kx = resample_ksize(W, OW), ky = resample_ksize(H, OH);
    if (horiz)
        hipLaunchKernelGGL(resample_coeffs_kernel, dim3((OW + 255) / 256), dim3(256), 0, stream, xtab, W, OW, kx);
    if (vert)
        hipLaunchKernelGGL(resample_coeffs_kernel, dim3((OH + 255) / 256), dim3(256), 0, stream, ytab, H, OH, ky);
    const uint8_t* mid = src;
    if (horiz) {
        uint8_t* out = vert ? tmp : dst;
        const long total = (long)B * H * OW * 3;
        hipLaunchKernelGGL(resample_pass_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                           src, out, xtab, total, W, OW, kx, 3L);
        mid = out;
    }
    if (vert) {
        const long total = (long)B * OH * OW * 3;
        hipLaunchKernelGGL(resample_pass_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                           mid, dst, ytab, total, H, OH, ky, (long)OW * 3);
    }
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

}  // namespace dir
