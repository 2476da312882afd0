// conv_patchlc.hip — layer1's 3x3 convolution (64 -> 64 channels, stride 1) with the filter RESIDENT in LDS and the work
// split by wave role (gfx950).
//
// conv_patch.hip's kernel for this layer (dirtorch/nets/backbones/resnet.py:58-59,74) streams the nine 8 KB filter taps
// through a 3-slot ring for every 256-pixel tile and has every wave both issue LDS-DMA and multiply: 0.16 ms per launch
// at batch 32 (3.3 TB/s of its 537 MB; 154.6 GFLOP).  Two findings of round 3 apply: on a
// memory-bound CU an LDS-DMA instruction holds its wave at issue, so waves that do both serialise the two phases; and
// L2 -> LDS traffic is not free next to HBM writes.  Here
//   * the whole filter (9 x 64 x 64 x 2 B = 72 KB) is loaded into LDS ONCE per persistent workgroup;
//   * the 10 x 34-pixel input patch of a tile (42.5 KB) is double-buffered: loader waves 4-6 fetch tile i + 1 while
//     consumer waves 0-3 (one per SIMD, two output rows x 32 pixels x 64 channels each, 64 accumulator registers)
//     multiply tile i straight from LDS - 144 MFMAs per wave per tile, no weight traffic at all;
//   * one fenced barrier per tile is the hand-off both ways (patch i landed / the other buffer is free);
//   * outputs leave straight from the accumulators: ReLU, pack, v_permlane32_swap pairs the two half-waves' 8-byte
//     pieces into 16-byte stores (CDNA guide T21).
// LDS: 73 728 (filter) + 2 x 44 032 (patches, whole 1 KiB DMA pieces) = 161 792 B of 163 840.  Swizzles, MFMA roles
// (A = weights, B = pixels), bias-initialised accumulators, tap and K order are those of conv_patch3x3_kernel: the
// outputs are bit-identical to it.
#include "dir_common.h"
#include "conv_igemm.h"

namespace dir {

static constexpr uint32_t kOOBl = 0x80000000u;

__device__ __forceinline__ void dma16l(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)lds, 16, voff, 0, 0, 0);
}

template <class DT>
__global__ void __launch_bounds__(512) conv_patch64_lc_kernel(const ConvArgs a) {
    constexpr int C = 64;
    constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2, PP = PH * PW;   // 340 patch pixels of 128 B
    constexpr int NPIECE = (PP + 7) / 8;          // 43 DMA pieces of 8 pixels (1 KiB)
    constexpr int PBUF = NPIECE * 1024;           // 44 032
    constexpr int WBYTES = 9 * C * 128;           // 73 728
    constexpr int NL = 3;                         // loader waves
    constexpr int LP = (NPIECE + NL - 1) / NL;    // pieces per loader wave (15)
    typedef typename DT::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;

    const int tiles_x = (a.OW + TW - 1) / TW;
    const int tiles_y = (a.OH + TH - 1) / TH;
    const int ntiles = a.B * tiles_y * tiles_x;
    const int first = blockIdx.x;
    if (first >= ntiles) return;
    const int my_tiles = (ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x;

    // ---- the filter, once: tap s = piece / 8, channels (piece % 8) * 8 .. + 7, 16-byte chunks swizzled with (n >> 1) & 7 ----
    {
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int q = k * 8 + wave;           // 72 pieces over 8 waves
            const int s = q >> 3, n = (q & 7) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((n >> 1) & 7);
            dma16l(rsrc_w, smem + q * 1024, (uint32_t)(((n * 9 + s) * C + chunk * 8) * 2));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_barrier();
    }

    if (wave == 7) {   // the eighth wave only keeps the barrier count
        for (int i = 0; i < my_tiles; ++i) ring_barrier();
        return;
    }
    if (wave >= 4) {
        // ================================ loaders ==============================================================
        const int lw = wave - 4;
        const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
        // (always_inline: left as a call, the lambda takes the by-value argument struct by reference and parks it in scratch)
        auto issue = [&](int tile, int buf) __attribute__((always_inline)) {
            int t = tile;
            const int tx = t % tiles_x;
            t /= tiles_x;
            const int ty = t % tiles_y;
            const int b = t / tiles_y;
            const int oy0 = ty * TH, ox0 = tx * TW;
            char* dst = smem + WBYTES + buf * PBUF;
#pragma unroll
            for (int k = 0; k < LP; ++k) {
                const int q = k * NL + lw;        // piece: patch pixels 8q .. 8q + 7
                if (q < NPIECE) {
                    const int p = q * 8 + (lane >> 3);
                    const int py = p / PW, px = p - py * PW;
                    const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
                    const bool ok = p < PP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    const int chunk = (lane & 7) ^ ((p >> 1) & 7);
                    const uint32_t v = ok ? (uint32_t)((((b * a.H + iy) * a.W + ix) * C + chunk * 8) * 2) : kOOBl;
                    dma16l(rsrc_x, dst + q * 1024, v);
                }
            }
        };
        issue(first, 0);
        for (int i = 0; i < my_tiles; ++i) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of patch i have landed
            ring_barrier();                                     // hand-off i: patch i complete; the consumers have left tile i - 1
            if (i + 1 < my_tiles) issue(first + (i + 1) * (int)gridDim.x, (i + 1) & 1);
        }
        return;
    }

    // ==================================== consumers =============================================================
    // bias of this lane's 32 accumulator channels (acc[i][j][4 g + e] = channel i*32 + 8 g + 4 lhi + e)
    f32x4_t bz[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) bz[i][g] = *(const DIR_GLOBAL f32x4_t*)(a.bias + i * 32 + 8 * g + 4 * lhi);
    const int wswz = (lane >> 1) & 7;
    int woffk[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) woffk[ks] = lrow * 128 + (((2 * ks + lhi) ^ wswz) << 4);

    Ovf<DT> ovf;
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = first + it * (int)gridDim.x;
        ring_barrier();   // hand-off `it` (see the loaders)
        const char* plane = smem + WBYTES + (it & 1) * PBUF;
        f32x16_t acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = bz[i][g][e];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const char* wst = smem + (r * 3 + s) * (C * 128);
                frag_t xf[2][4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int p = (wave * 2 + j + r) * PW + s + lrow;   // patch pixel read by this lane
                    const int swz = (p >> 1) & 7;
                    const char* row = plane + p * 128;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) xf[j][ks] = *(const frag_t*)(row + (((2 * ks + lhi) ^ swz) << 4));
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    frag_t wf[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) wf[i] = *(const frag_t*)(wst + i * 4096 + woffk[ks]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = DT::mfma32(wf[i], xf[j][ks], acc[i][j]);
                }
            }
        // ---- ReLU, pack; v_permlane32_swap on the pair (g, g + 1) leaves lanes 0-31 with channels 8 g .. 8 g + 7 and lanes
        //      32-63 with the next eight: one 16-byte store each ------------------------------------------------
        int t = tile;
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y;
        const int b = t / tiles_y;
        const int ox = tx * TW + lrow;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oy = ty * TH + wave * 2 + j;
            const bool ok = oy < a.OH && ox < a.OW;
            uint16_t* yrow = a.y + ((size_t)((b * a.OH + (ok ? oy : 0)) * a.OW + (ok ? ox : 0)) * C + lhi * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    uint32_t q2[2][2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = acc[i][j][4 * (2 * h + q) + e];
                            v[e] = a.relu ? fmaxf(x, 0.f) : x;
                        }
                        q2[q][0] = DT::pack(v[0], v[1]);
                        q2[q][1] = DT::pack(v[2], v[3]);
                    }
                    const auto r0 = __builtin_amdgcn_permlane32_swap(q2[0][0], q2[1][0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(q2[0][1], q2[1][1], false, false);
                    const u32x4_t ov = {r0[0], r1[0], r0[1], r1[1]};
                    if (ok) {
                        gstore16(yrow + i * 32 + h * 16, ov);
                        ovf.see(ov);
                    }
                }
        }
    }
    ovf.flush(a.ovf);
}

bool conv_patch64_lc_admissible(const ConvArgs& a) {
    return a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.H == a.OH && a.W == a.OW && a.Cin == 64 &&
           a.Cout == 64 && a.res == nullptr && (size_t)a.B * a.H * a.W * 64 * 2 < (1ull << 31);
}

template <class DT>
static hipError_t launch_patch_lc(const ConvArgs& a, hipStream_t stream) {
    constexpr int LDS = 9 * 64 * 128 + 2 * 43 * 1024;   // filter + two patch buffers = 161 792
    static_assert(LDS <= 160 * 1024, "LDS map");
    auto kern = conv_patch64_lc_kernel<DT>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    const long tiles = (long)a.B * ((a.OH + 7) / 8) * ((a.OW + 31) / 32);
    const int ncu = cu_count();
    const int grid = tiles < ncu ? (int)tiles : ncu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, b);
    return hipGetLastError();
}

hipError_t conv_patch64_lc_launch(const ConvArgs& a, int dtype, hipStream_t stream) {
    return dtype == DIR_BF16 ? launch_patch_lc<BF16>(a, stream) : launch_patch_lc<FP16>(a, stream);
}

}  // namespace dir
