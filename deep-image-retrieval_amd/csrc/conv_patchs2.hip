// conv_patchs2.hip — 3x3 STRIDE-2 convolutions (pad 1) from an LDS-resident patch: conv2 of the first block of layers 2-4
// (dirtorch/nets/backbones/resnet.py:58-59 with stride 2, :73-75), persistent workgroups with loader / consumer wave roles
// (gfx950, round 6).
//
// Those three launches ran on the generic implicit-GEMM tiles (conv_igemm.hip), whose LDS-DMA gather fetches every input
// pixel 9/4 times from L2 and every weight once per 256 pixels: layer2.0's 128 -> 128 conv moved 1.8 GB through the CUs'
// memory pipes for 0.67 GB of tensors (0.245 ms, 0.34 of the HBM roof).  Here
//   * the tile is 8 x 32 output pixels x 128 channels; its 17 x 65-pixel input patch is staged ONCE, one 32-channel plane
//     at a time (70 KB), two plane buffers: plane g + 1 of the workgroup's whole tile sequence is in flight while plane g
//     is multiplied - the next tile's first plane lands under this tile's last MFMAs and its stores;
//   * the patch's EVEN and ODD input columns are separate runs of a plane row (slots 0-32 / 33-64), so the 32 lanes of a
//     pixel fragment - output columns ox .. ox + 31 of tap s = input columns 2 ox + s - read CONSECUTIVE 64-byte slots
//     (even run at +0, odd run at +0, even run at +1): the conflict-free pattern of the stride-1 kernels (chunks XOR-swizzled
//     by the slot's index in its row, (c >> 2) & 3), where the natural layout would put every lane pair on the same banks;
//   * weights never touch LDS: a consumer wave owns 32 output channels x 4 output rows, and loads its weight fragments
//     (one filter row of one plane: 6 x 16 bytes per lane) from L2 straight into registers one filter row ahead of the
//     MFMAs that use them - from a FRAGMENT-ORDERED copy of the filter (pack_patchs2_kernel: the engine keeps one per
//     strided layer since finalize(), the per-op entry point packs into stream-ordered scratch), so that a wave-level load
//     is one contiguous KB.  (First build, from the [Cout][3][3][Cin] layout: 32 segments of 32 bytes per instruction,
//     ~260 cycles each in the CU's texture path - 0.6 ms per launch, 18 us per plane whatever the layer.)
//   * outputs leave straight from the accumulators (ReLU, pack, v_permlane32_swap -> one 16-byte store per lane).
// Twelve waves: 0-7 multiply (wave = channel tile w & 3, row half w >> 2; 72 MFMAs per plane, 64 accumulator registers),
// 8-11 only issue LDS-DMA (18 x 1 KB pieces per plane each) and wait for it; ONE workgroup barrier per plane is the
// hand-off in both directions.  LDS: 2 x 70 KB.
#include "dir_common.h"
#include "conv_igemm.h"

namespace dir {

static constexpr uint32_t kOOBs2 = 0x80000000u;

__device__ __forceinline__ void dma16s2(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)lds, 16, voff, soff, 0, 0);
}

template <class DT>
__global__ void __launch_bounds__(768) conv_patch3x3s2_kernel(const ConvArgs a) {
    constexpr int TH = 8, TW = 32, PH = 2 * TH + 1, PW = 2 * TW + 1, PP = PH * PW;   // 17 x 65 = 1105 patch pixels
    constexpr int NEV = TW + 1;                           // even-column run of a row: slots 0 .. 32; odd columns: 33 .. 64
    constexpr int NPIECE = (PP + 15) / 16;                // 70 DMA pieces of 16 slots x 64 B
    constexpr int NL = 4, LP = (NPIECE + NL - 1) / NL;    // 18 pieces per loader wave and plane
    constexpr int PBUF = NPIECE * 1024;                   // 71 680
    constexpr int BN = 128;
    typedef typename DT::frag_t frag_t;
    static_assert(2 * PBUF <= 160 * 1024, "LDS map");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;

    const int NQ = a.Cin / 32;                 // planes per tile (even: Cin % 64 == 0)
    const int tiles_n = a.Cout / BN;
    const int tiles_x = (a.OW + TW - 1) / TW;
    const int tiles_y = (a.OH + TH - 1) / TH;
    const int tiles = a.B * tiles_y * tiles_x * tiles_n;
    const int G = (int)gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);   // neighbours in the tile order (channel tiles of one patch first) share an L2
    const int my_tiles = (tiles - first + G - 1) / G;

    if (wave >= 8) {
        // ================================ loaders ==============================================================================
        const int lw = wave - 8;
        const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
        // this lane's slot of piece k (the same in every plane of every tile): patch row / input column offset, chunk
        int ppos[LP];
#pragma unroll
        for (int k = 0; k < LP; ++k) {
            const int piece = k * NL + lw;
            const int p = piece * 16 + (lane >> 2);
            const int py = p / PW, c = p - py * PW;
            const int px = c < NEV ? 2 * c : 2 * (c - NEV) + 1;
            const int chunk = (lane & 3) ^ ((c >> 2) & 3);
            ppos[k] = (piece < NPIECE && p < PP) ? (py << 16) | (px << 4) | chunk : -1;
        }
        uint32_t pvoff[LP];
        auto locate = [&](int tile) __attribute__((always_inline)) {
            int t = tile / tiles_n;
            const int tx = t % tiles_x;
            t /= tiles_x;
            const int ty = t % tiles_y;
            const int b = t / tiles_y;
            const int iy0 = 2 * ty * TH - 1, ix0 = 2 * tx * TW - 1;
#pragma unroll
            for (int k = 0; k < LP; ++k) {
                const int iy = iy0 + (ppos[k] >> 16), ix = ix0 + ((ppos[k] >> 4) & 0xfff);
                const bool ok = ppos[k] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                pvoff[k] = ok ? (uint32_t)((((b * a.H + iy) * a.W + ix) * a.Cin + (ppos[k] & 15) * 8) * 2) : kOOBs2;
            }
        };
        auto issue = [&](int q, int buf) __attribute__((always_inline)) {
            char* dst = smem + buf * PBUF;
#pragma unroll
            for (int k = 0; k < LP; ++k)
                if (k * NL + lw < NPIECE) dma16s2(rsrc_x, dst + (k * NL + lw) * 1024, pvoff[k], q * 64);
        };
        const int NG = my_tiles * NQ;
        int it = 0, q = 0;
        locate(first);
        issue(0, 0);
        for (int g = 0; g < NG; ++g) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of plane g have landed
            ring_barrier();                                     // hand-off g: plane g complete; the consumers have left plane g - 1
            if (g + 1 < NG) {
                if (++q == NQ) {
                    q = 0;
                    ++it;
                    locate(first + it * G);
                }
                issue(q, (g + 1) & 1);
            }
        }
        return;
    }

    // ==================================== consumers ================================================================================
    const int ct = wave & 3, rh = wave >> 2;
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_s2, 0, a.w_bytes, 0x00020000);
    const uint32_t wvoff = (uint32_t)(lane * 16);
    frag_t W[2][3][2];   // two sets (this filter row / the next one) x tap x K half
    auto load_w = [&](frag_t (&w)[3][2], int tile_n, int q, int r) __attribute__((always_inline)) {
        const int blk = ((((tile_n * NQ + q) * 3 + r) * 4 + ct) * 6) * 1024;   // pack_patchs2_kernel's order
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                w[s][kk] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, wvoff, blk + (s * 2 + kk) * 1024, 0));
    };
    // fragment of output row j (of this wave's four), filter row r, tap s: slot (2 (4 rh + j) + r) * 65 + c, c = {0, 33, 1}[s] + lrow;
    // its 16-byte chunks are swizzled by the slot's index IN ITS ROW, (c >> 2) & 3, so that a lane has one address per (tap, K half)
    // and the rows are immediate offsets
    const char* xbase[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int c = (s == 1 ? NEV : (s >> 1)) + lrow;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xbase[s][kk] = smem + ((2 * 4 * rh) * PW + c) * 64 + (((2 * kk + lhi) ^ ((c >> 2) & 3)) << 4);
    }

    Ovf<DT> ovf;
    load_w(W[0], first % tiles_n, 0, 0);
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = first + it * G;
        const int tile_n = tile % tiles_n;
        const int tn_next = it + 1 < my_tiles ? (tile + G) % tiles_n : tile_n;
        f32x16_t acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_t b4 = *(const DIR_GLOBAL f32x4_t*)(a.bias + tile_n * BN + ct * 32 + 8 * g + 4 * lhi);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][4 * g + e] = b4[e];
        }
        auto plane_step = [&](int q, auto par) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par)::value;
            ring_barrier();   // hand-off (see the loaders); NQ is even, so plane q of any tile sits in buffer q & 1 = PAR
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int cur = (PAR + r) & 1;
                // the next filter row's fragments (the next plane's, the next tile's) go out before this row's MFMAs
                if (r < 2)
                    load_w(W[cur ^ 1], tile_n, q, r + 1);
                else if (q + 1 < NQ)
                    load_w(W[cur ^ 1], tile_n, q + 1, 0);
                else
                    load_w(W[cur ^ 1], tn_next, 0, 0);
                // (compiler fence: rows 2 j + r coincide for (j, r + 2) and (j + 1, r), and kept across filter rows for re-use those
                // fragments cost more registers than the 168 of a twelve-wave workgroup hold - every filter row reads its own)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        frag_t xf[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) xf[j] = *(const frag_t*)(xbase[s][kk] + PAR * PBUF + (2 * j + r) * (PW * 64));
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = DT::mfma32(W[cur][s][kk], xf[j], acc[j]);
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this plane's LDS reads retired before the next barrier
        };
        for (int q = 0; q < NQ; q += 2) {
            plane_step(q, std::integral_constant<int, 0>{});
            plane_step(q + 1, std::integral_constant<int, 1>{});
        }
        // ---- ReLU, pack; v_permlane32_swap on the pair (g, g + 1) leaves lanes 0-31 with channels 8 g .. 8 g + 7 and lanes
        //      32-63 with the next eight: one 16-byte store each ------------------------------------------------------------------
        int t = tile / tiles_n;
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y;
        const int b = t / tiles_y;
        const int ox = tx * TW + lrow;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = ty * TH + rh * 4 + j;
            const bool ok = oy < a.OH && ox < a.OW;
            uint16_t* yrow = a.y + ((size_t)((b * a.OH + (ok ? oy : 0)) * a.OW + (ok ? ox : 0)) * a.Cout + tile_n * BN + ct * 32 + lhi * 8);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t q2[2][2];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = acc[j][4 * (2 * h + qq) + e];
                        v[e] = a.relu ? fmaxf(x, 0.f) : x;
                    }
                    q2[qq][0] = DT::pack(v[0], v[1]);
                    q2[qq][1] = DT::pack(v[2], v[3]);
                }
                const auto r0 = __builtin_amdgcn_permlane32_swap(q2[0][0], q2[1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(q2[0][1], q2[1][1], false, false);
                const u32x4_t ov = {r0[0], r1[0], r0[1], r1[1]};
                if (ok) {
                    gstore16(yrow + h * 16, ov);
                    ovf.see(ov);
                }
            }
        }
    }
    ovf.flush(a.ovf);
}

bool conv_patch3x3s2_admissible(const ConvArgs& a) {
    return a.R == 3 && a.S == 3 && a.stride == 2 && a.pad == 1 && a.OH == (a.H - 1) / 2 + 1 && a.OW == (a.W - 1) / 2 + 1 &&
           a.Cin % 64 == 0 && a.Cin >= 64 && a.Cout % 128 == 0 && a.res == nullptr && a.ksplit <= 1 &&
           (size_t)a.B * a.H * a.W * a.Cin * 2 < (1ull << 31) && (size_t)a.Cout * a.Ktot * 2 < (1ull << 31);
}

// The filter in the order the consumers read it: 16-byte piece ((((tile_n * NQ + q) * 3 + r) * 4 + ct) * 6 + s * 2 + kk) * 64 + lane
// = w[tile_n * 128 + ct * 32 + (lane & 31)][r][s][q * 32 + kk * 16 + (lane >> 5) * 8 .. + 8]  (same bytes, same size).
__global__ void __launch_bounds__(256) pack_patchs2_kernel(const uint16_t* w, uint16_t* out, int Cout, int Cin) {
    const int NQ = Cin / 32;
    const long pieces = (long)Cout * 9 * Cin / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (long)gridDim.x * 256) {
        long t = i;
        const int lane = (int)(t & 63);
        t >>= 6;
        const int sk = (int)(t % 6);
        t /= 6;
        const int ct = (int)(t & 3);
        t >>= 2;
        const int r = (int)(t % 3);
        t /= 3;
        const int q = (int)(t % NQ);
        const int tn = (int)(t / NQ);
        const int s = sk >> 1, kk = sk & 1;
        const size_t src = ((size_t)((tn * 128 + ct * 32 + (lane & 31)) * 9 + r * 3 + s)) * Cin + q * 32 + kk * 16 + (lane >> 5) * 8;
        gstore16(out + i * 8, gload16(w + src));
    }
}

hipError_t conv_patch3x3s2_pack(const uint16_t* w, uint16_t* out, int Cout, int Cin, hipStream_t stream) {
    const long pieces = (long)Cout * 9 * Cin / 8;
    const int grid = (int)((pieces + 255) / 256 < 1024 ? (pieces + 255) / 256 : 1024);
    hipLaunchKernelGGL(pack_patchs2_kernel, dim3(grid), dim3(256), 0, stream, w, out, Cout, Cin);
    return hipGetLastError();
}

template <class DT>
static hipError_t launch_patch_s2(const ConvArgs& a, hipStream_t stream) {
    constexpr int LDS = 2 * 70 * 1024;
    auto kern = conv_patch3x3s2_kernel<DT>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    void* scratch = nullptr;
    if (!b.w_s2) {   // the per-op entry point (dir_conv_bn_act): no packed copy kept anywhere - stream-ordered scratch for this launch
        if (hipError_t e = hipMallocAsync(&scratch, b.w_bytes, stream); e != hipSuccess) return e;
        if (hipError_t e = conv_patch3x3s2_pack(a.w, (uint16_t*)scratch, a.Cout, a.Cin, stream); e != hipSuccess) {
            (void)hipFreeAsync(scratch, stream);
            return e;
        }
        b.w_s2 = (const uint16_t*)scratch;
    }
    const long tiles = (long)a.B * ((a.OH + 7) / 8) * ((a.OW + 31) / 32) * (a.Cout / 128);
    const int ncu = cu_count();
    const int grid = tiles < ncu ? (int)tiles : ncu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(768), LDS, stream, b);
    hipError_t e = hipGetLastError();
    if (scratch) {
        const hipError_t f = hipFreeAsync(scratch, stream);
        if (e == hipSuccess) e = f;
    }
    return e;
}

hipError_t conv_patch3x3s2_launch(const ConvArgs& a, int dtype, hipStream_t stream) {
    return dtype == DIR_BF16 ? launch_patch_s2<BF16>(a, stream) : launch_patch_s2<FP16>(a, stream);
}

}  // namespace dir
