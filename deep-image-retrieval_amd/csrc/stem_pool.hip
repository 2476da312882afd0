// stem_pool.hip — the ResNet stem in one kernel: 7x7 s2 conv + BN + ReLU + 3x3 s2 max-pool
// (dirtorch/nets/backbones/resnet.py:115-119,158-161), gfx950.
//
// Unfused, the stem writes B x (H/2 x W/2) x 64 activations (33.5 MB per 1024^2 image) only for the
// max-pool to read them back and keep a quarter.  Here a workgroup produces a 3 x 15 tile of POOLED
// pixels: it loads the 11 x 35 pixel patch of the space-to-depth image (see prep_input) and the
// whole 32 KiB stem filter into LDS once, computes the 8 x 32 conv outputs that the pool windows
// touch (7 x 31 used) with 32x32x16 MFMAs straight from the patch, keeps them in LDS, and writes
// only the pooled tile.  HBM traffic: s2d image once (8.4 MB/img) + pooled map (8.4 MB/img).
//
// GEMM view per K-step R (filter row of the 4x4 s2d filter): the 64 K-elements of output pixel
// (oy, ox) are the 16 channels of patch pixels (oy+R, ox..ox+3); k-substep ks = pixel ox+ks.  The
// patch is stored as two planes (channels 0-7 / 8-15, 16 B per pixel each) so that a fragment read
// (32 consecutive pixels, one plane per half-wave) is one contiguous 512-byte run.
// Out-of-image conv outputs are stored as 0, which is exact for the max: every pool window holds
// at least one real post-ReLU (>= 0) value.
#include "dir_common.h"
#include "conv_igemm.h"

namespace dir {

static constexpr uint32_t kOOBs = 0x80000000u;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;

__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)lds, 16, voff, soff, 0, 0);
}

struct StemPoolArgs {
    const uint16_t* x;   // s2d image [B, H2, W2, 16]
    const uint16_t* w;   // [64][4][4][16]
    const float* bias;   // [64]
    uint16_t* y;         // pooled [B, PH, PW, 64]
    int B, H2, W2, OH, OW, PH, PW;
    uint32_t x_bytes;
    int* ovf;            // fp16 overflow word (dir_common.h Ovf), or nullptr
};

template <class DT>
__global__ void __launch_bounds__(256) stem_pool_kernel(const StemPoolArgs a) {
    constexpr int PTH = 3, PTW = 15;         // pooled tile
    constexpr int TH = 8, TW = 32;           // conv tile (rows 2*ph0-1 .., cols 2*pw0-1 ..)
    constexpr int QW = TW + 3;               // patch width (35); patch height TH + 3 = 11
    constexpr int QP = (TH + 3) * QW;        // 385 patch pixels
    constexpr int PLANE = 512 * 16;          // one channel-half plane, padded to 2 DMA instructions
    constexpr int WOFF = 2 * PLANE;          // filter after the patch: 4 x [64][64] swizzled slices
    static_assert(TH * TW * 128 <= 2 * PLANE + 4 * 8192, "the conv-output tile aliases patch + filter");
    typedef typename DT::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31;
    const int lhi = lane >> 5;

    const int tiles_x = (a.PW + PTW - 1) / PTW;
    const int tiles_y = (a.PH + PTH - 1) / PTH;
    int wg = blockIdx.x;
    const int tx = wg % tiles_x;
    wg /= tiles_x;
    const int ty = wg % tiles_y;
    const int b = wg / tiles_y;
    const int ph0 = ty * PTH, pw0 = tx * PTW;
    const int oy0 = 2 * ph0 - 1, ox0 = 2 * pw0 - 1;  // conv-output origin of the tile

    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 64 * 256 * 2, 0x00020000);

    // ---- patch (2 planes x 512 slots) and the whole filter ------------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int P = i * 256 + tid;
        const int plane = P >> 9, p = P & 511;
        const int py = p / QW, px = p - py * QW;
        const int iy = oy0 - 2 + py, ix = ox0 - 2 + px;  // s2d pixel (conv pad 2 on top/left)
        const bool ok = p < QP && (unsigned)iy < (unsigned)a.H2 && (unsigned)ix < (unsigned)a.W2;
        const uint32_t v = ok ? (uint32_t)((((b * a.H2 + iy) * a.W2 + ix) * 16 + plane * 8) * 2) : kOOBs;
        dma16s(rsrc_x, smem + (i * 256 + wave * 64) * 16, v, 0);
    }
    const int srcchunk = (tid & 7) ^ ((tid >> 4) & 7);
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = i * 32 + (tid >> 3);
            dma16s(rsrc_w, smem + WOFF + R * 8192 + (i * 256 + wave * 64) * 16,
                   (uint32_t)((n * 256 + srcchunk * 8) * 2), R * 128);
        }

    // ---- accumulators start at the bias: 2 channel tiles x 2 conv rows per wave --------------------
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_t b4 = *(const DIR_GLOBAL f32x4_t*)(a.bias + i * 32 + 8 * g + 4 * lhi);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = b4[e];
        }
    const int wswz = (lane >> 1) & 7;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int R = 0; R < 4; ++R) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            frag_t wf[2], xf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                wf[i] = *(const frag_t*)(smem + WOFF + R * 8192 + (i * 32 + lrow) * 128 +
                                         (((2 * ks + lhi) ^ wswz) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = (wave * 2 + j + R) * QW + lrow + ks;  // patch pixel (oy+R, ox+ks)
                xf[j] = *(const frag_t*)(smem + lhi * PLANE + p * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = DT::mfma32(wf[i], xf[j], acc[i][j]);
        }
    }
    __syncthreads();  // patch and filter are dead: the conv tile takes their place

    // ---- ReLU, convert, keep the conv tile in LDS: pixel-major rows of 128 B, 16-byte chunk index
    //      XOR-ed with (x & 7) so that neither these 8-byte writes nor the pooling reads pile up ------
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tyy = wave * 2 + j;
        const int oy = oy0 + tyy, ox = ox0 + lrow;
        const uint32_t inmask = ((unsigned)oy < (unsigned)a.OH && (unsigned)ox < (unsigned)a.OW) ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];   // ReLU as a signed-integer max on the bit pattern: negatives, -0 (and -NaN) -> +0
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[i][j][4 * g + e];   // (a bit_cast straight from the vector element reads element 0)
                    v[e] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, t), 0));
                }
                u32x2_t pk = {DT::pack(v[0], v[1]) & inmask, DT::pack(v[2], v[3]) & inmask};
                const int c = 4 * i + g;  // 16-byte chunk (8 channels) of this pixel
                *(u32x2_t*)(smem + (tyy * TW + lrow) * 128 + ((c ^ (lrow & 7)) << 4) + lhi * 8) = pk;
            }
    }
    __syncthreads();

    // ---- 3x3 stride-2 max over the tile, 16 bytes (8 channels) per work item -------------------------
    Ovf<DT> ovf;
    for (int it = tid; it < PTH * PTW * 8; it += 256) {
        const int c = it & 7;
        const int pp = it >> 3;
        const int py = pp / PTW, px = pp - py * PTW;
        const int ph = ph0 + py, pw = pw0 + px;
        if (ph >= a.PH || pw >= a.PW) continue;
        // the tile holds post-ReLU values (>= +0): for those the 16-bit patterns of bf16 / fp16 order like unsigned
        // integers, so the max is v_pk_max_u16 on the packed words - 4 VALU per 8 channels per tap, not 16 + 8
        u16x8_t best = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int yy = 2 * py + dy, xx = 2 * px + dx;
                best = __builtin_elementwise_max(
                    best, *(const u16x8_t*)(smem + (yy * TW + xx) * 128 + ((c ^ (xx & 7)) << 4)));
            }
        gstore16(a.y + ((size_t)(b * a.PH + ph) * a.PW + pw) * 64 + c * 8, __builtin_bit_cast(u32x4_t, best));
        ovf.see(__builtin_bit_cast(u32x4_t, best));
    }
    ovf.flush(a.ovf);
}

// ---- persistent form ---------------------------------------------------------------------------------------
// The one-tile-per-workgroup kernel above re-loads the whole 32 KiB filter for every 3 x 15 pooled tile (44 KB of
// L2 -> LDS traffic in front of 0.9 us of MFMA work) and runs its phases - load, MFMA, conv tile to LDS, pool, store
// - strictly one after the other: MfmaUtil 0.46, 0.43 ms at batch 32.  Here two workgroups per CU walk strided lists
// of tiles with the FILTER IN REGISTERS (32 fragments = 128 VGPRs per wave, fetched once per workgroup straight in
// MFMA operand layout), two patch buffers (the next tile's patch is in flight while this one is multiplied and
// pooled) and the conv tile: 64 KiB of LDS.  Per tile: 16 KB of patch in, 5.8 KB of pooled pixels out, two barriers.
template <class DT>
__global__ void __launch_bounds__(256, 2) stem_pool_persist_kernel(const StemPoolArgs a) {
    constexpr int PTH = 3, PTW = 15;
    constexpr int TH = 8, TW = 32;
    constexpr int QW = TW + 3;
    constexpr int QP = (TH + 3) * QW;
    constexpr int PLANE = 512 * 16;
    constexpr int PATCH = 2 * PLANE;          // 16 KiB
    constexpr int TILE_OFF = 2 * PATCH;       // conv tile behind the two patch buffers
    typedef typename DT::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31;
    const int lhi = lane >> 5;

    const int tiles_x = (a.PW + PTW - 1) / PTW;
    const int tiles_y = (a.PH + PTH - 1) / PTH;
    const int ntiles = a.B * tiles_y * tiles_x;

    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);

    // filter fragments: A operand of step (R, ks) for channel tile i = 8 K-elements of row i*32 + lrow
    frag_t wf[4][4][2];
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                wf[R][ks][i] = __builtin_bit_cast(
                    frag_t, gload16(a.w + (i * 32 + lrow) * 256 + R * 64 + ks * 16 + lhi * 8));
    // the bias lives in LDS (256 B behind the conv tile) and is re-read at the top of every tile: the filter
    // already takes half of the 256 registers a wave may have at two workgroups per CU
    float* lbias = (float*)(smem + TILE_OFF + TH * TW * 128);
    if (tid < 64) lbias[tid] = a.bias[tid];

    auto issue_patch = [&](int tile, char* dst) {
        int wg = tile;
        const int tx = wg % tiles_x;
        wg /= tiles_x;
        const int ty = wg % tiles_y;
        const int b = wg / tiles_y;
        const int oy0 = 2 * ty * PTH - 1, ox0 = 2 * tx * PTW - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int P = i * 256 + tid;
            const int plane = P >> 9, p = P & 511;
            const int py = p / QW, px = p - py * QW;
            const int iy = oy0 - 2 + py, ix = ox0 - 2 + px;
            const bool ok = p < QP && (unsigned)iy < (unsigned)a.H2 && (unsigned)ix < (unsigned)a.W2;
            const uint32_t v = ok ? (uint32_t)((((b * a.H2 + iy) * a.W2 + ix) * 16 + plane * 8) * 2) : kOOBs;
            dma16s(rsrc_x, dst + (i * 256 + wave * 64) * 16, v, 0);
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    Ovf<DT> ovf;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // filter in registers, bias in LDS, before the counted waits start
    issue_patch(tile, smem);
    int cur = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        char* pbuf = smem + cur * PATCH;
        // the other buffer was last read by the MFMA phase of the previous tile, which every wave left before
        // the barrier in front of that tile's pooling
        if (next < ntiles) {
            issue_patch(next, smem + (cur ^ 1) * PATCH);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // this tile's patch; the 4 newest ops may fly
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ring_barrier();   // patch landed everywhere; everyone is done pooling the previous tile

        f32x16_t acc[2][2];   // start at the bias, like every conv kernel of this library (same fp32 order)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t b4 = *(const f32x4_t*)(lbias + i * 32 + 8 * g + 4 * lhi);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = b4[e];
            }
#pragma unroll
        for (int R = 0; R < 4; ++R)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                frag_t xf[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int p = (wave * 2 + j + R) * QW + lrow + ks;
                    xf[j] = *(const frag_t*)(pbuf + lhi * PLANE + p * 16);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = DT::mfma32(wf[R][ks][i], xf[j], acc[i][j]);
            }

        int wg = tile;
        const int tx = wg % tiles_x;
        wg /= tiles_x;
        const int ty = wg % tiles_y;
        const int b = wg / tiles_y;
        const int ph0 = ty * PTH, pw0 = tx * PTW;
        const int oy0 = 2 * ph0 - 1, ox0 = 2 * pw0 - 1;
        char* ctile = smem + TILE_OFF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tyy = wave * 2 + j;
            const int oy = oy0 + tyy, ox = ox0 + lrow;
            const uint32_t inmask = ((unsigned)oy < (unsigned)a.OH && (unsigned)ox < (unsigned)a.OW) ? 0xffffffffu : 0u;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = acc[i][j][4 * g + e];
                        v[e] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, t), 0));
                    }
                    u32x2_t pk = {DT::pack(v[0], v[1]) & inmask, DT::pack(v[2], v[3]) & inmask};
                    const int c = 4 * i + g;
                    *(u32x2_t*)(ctile + (tyy * TW + lrow) * 128 + ((c ^ (lrow & 7)) << 4) + lhi * 8) = pk;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ring_barrier();   // conv tile complete; every wave is past its patch reads

        for (int it = tid; it < PTH * PTW * 8; it += 256) {
            const int c = it & 7;
            const int pp = it >> 3;
            const int py = pp / PTW, px = pp - py * PTW;
            const int ph = ph0 + py, pw = pw0 + px;
            if (ph >= a.PH || pw >= a.PW) continue;
            u16x8_t best = {0, 0, 0, 0, 0, 0, 0, 0};   // packed unsigned max: see the one-tile kernel above
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int yy = 2 * py + dy, xx = 2 * px + dx;
                    best = __builtin_elementwise_max(
                        best, *(const u16x8_t*)(ctile + (yy * TW + xx) * 128 + ((c ^ (xx & 7)) << 4)));
                }
            gstore16(a.y + ((size_t)(b * a.PH + ph) * a.PW + pw) * 64 + c * 8, __builtin_bit_cast(u32x4_t, best));
            ovf.see(__builtin_bit_cast(u32x4_t, best));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // pooling reads retired before the next tile's barrier
        cur ^= 1;
    }
    ovf.flush(a.ovf);
}

int stem_pool_launch(const void* s2d, const void* w, const float* bias, void* y, int B, int H2, int W2,
                     int OH, int OW, int dtype, hipStream_t stream, int* ovf) {
    if ((size_t)B * H2 * W2 * 32 >= (1ull << 31))
        return fail(DIR_ERR_INVALID, "stem_pool: input exceeds 2^31 bytes; lower the batch");
    StemPoolArgs a;
    a.x = (const uint16_t*)s2d;
    a.w = (const uint16_t*)w;
    a.bias = bias;
    a.y = (uint16_t*)y;
    a.B = B; a.H2 = H2; a.W2 = W2; a.OH = OH; a.OW = OW;
    a.PH = (OH - 1) / 2 + 1;
    a.PW = (OW - 1) / 2 + 1;
    a.x_bytes = (uint32_t)((size_t)B * H2 * W2 * 32);
    a.ovf = ovf;
    const long blocks = (long)B * ((a.PH + 2) / 3) * ((a.PW + 14) / 15);
    if (dtype != DIR_BF16 && dtype != DIR_FP16) return fail(DIR_ERR_INVALID, "stem_pool: bad dtype");
    const bool v1 = env().stem_v1;   // A/B and bisecting (dir_reload_env after flipping it)
    if (!v1) {
        constexpr int LDSP = 2 * 2 * 512 * 16 + 8 * 32 * 128 + 256;   // two patch buffers + the conv tile + bias
        const long grid = blocks < 2L * cu_count() ? blocks : 2L * cu_count();
        if (dtype == DIR_BF16)
            hipLaunchKernelGGL(stem_pool_persist_kernel<BF16>, dim3((unsigned)grid), dim3(256), LDSP, stream, a);
        else
            hipLaunchKernelGGL(stem_pool_persist_kernel<FP16>, dim3((unsigned)grid), dim3(256), LDSP, stream, a);
        DIR_HIP_CHECK(hipGetLastError());
        return DIR_OK;
    }
    constexpr int LDS = 2 * 512 * 16 + 4 * 8192;  // 48 KiB >= the 32 KiB conv tile
    if (dtype == DIR_BF16)
        hipLaunchKernelGGL(stem_pool_kernel<BF16>, dim3((unsigned)blocks), dim3(256), LDS, stream, a);
    else if (dtype == DIR_FP16)
        hipLaunchKernelGGL(stem_pool_kernel<FP16>, dim3((unsigned)blocks), dim3(256), LDS, stream, a);
    else
        return fail(DIR_ERR_INVALID, "stem_pool: bad dtype");
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

}  // namespace dir
