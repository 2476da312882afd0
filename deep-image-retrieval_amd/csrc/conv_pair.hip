// conv_pair.hip — the PAIRED-fp16 kernels of the DIR_FP16P precision mode (gfx950).
//
// The reference computes every layer in fp32 (dirtorch/nets/backbones/resnet.py:67-87,115-119).  A 16-bit engine
// rounds weights and activations to 11 significant bits, and on a well-conditioned (BatchNorm-calibrated) network that
// rounding alone costs 1.3e-4 of descriptor cosine - over the 1e-4 the north star allows.  Where that error is MADE is
// very uneven (tests/precision_decomposition.py, the fp32 oracle with single storage points rounded): the input image,
// the stem and layer1 account for ~94 % of it - perturbations made there travel through every later BatchNorm-scaled
// layer - while layer3 / layer4, 60 % of the arithmetic, contribute under 1e-6.  DIR_FP16P therefore keeps the fast
// fp16 kernels for layers 2-4 and runs the image, the stem and layer1 on PAIRS of fp16 values:
//     v  ~  hi + lo,   hi = fp16(v),  lo = fp16(v - hi)          (~22 significant bits)
// stored as two planes of the ordinary layout (NHWC activations, [Cout][R][S][Cin] weights), and every product as
// three fp16 MFMAs into the one fp32 accumulator:   w.x  ~  wh.xh + wh.xl + wl.xh   (wl.xl ~ 2^-22 is dropped).
// The matrix pipe has room for that in exactly these layers: they are HBM-bound (K = 64 ... 576 per output).
//
//   conv_pair_kernel        any conv + folded BatchNorm (+ residual pair) (+ ReLU) -> pair (or single-plane) output
//   stem_pool_pair_kernel   7x7 s2 conv + BN + ReLU + 3x3 s2 max-pool on the space-to-depth image pair, pooled in fp32
//   prep_input_pair_kernel  image (fp32 NCHW normalised | uint8 NHWC raw) -> space-to-depth NHWC16 pair
// Structure (LDS-DMA through buffer descriptors, XOR-swizzled rows, swapped MFMA roles, bias-initialised accumulators,
// fp32 LDS staging for 16-byte stores) is conv_igemm.hip's; only what the pairs change is new.
#include "dir_common.h"
#include "conv_igemm.h"
#include "pointwise.h"

// Timing-only experiment builds of stem_pool_pair_persist_kernel (scripts/exp_abl.sh conv_pair DIR_STEMP_ABL <bits>): 1 = no MFMAs /
// fragment reads, 2 = no pooling phase (conv tile written, never read; no stores), 4 = no patch DMA.  NOT valid results.
#ifndef DIR_STEMP_ABL
#define DIR_STEMP_ABL 0
#endif

namespace dir {

static constexpr uint32_t kOOBp = 0x80000000u;

__device__ __forceinline__ void dma16p(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ uint32_t fast_div_p(uint32_t n, uint32_t mul, uint32_t shr) {
    return mul ? (__umulhi(n, mul) >> shr) : n;
}

// fp32 pair (a, b) -> packed hi word and packed lo word: hi = fp16(v), lo = fp16(v - hi)
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = FP16::pack(a, b);
    float ha, hb;
    FP16::unpack(hi, ha, hb);
    lo = FP16::pack(a - ha, b - hb);
}

// ---- convolution ----------------------------------------------------------------------------------------------------
// K-step = 32 input channels of one filter tap; a stage = Xh [BM][32] (+ Xl) + Wh [BN][32] + Wl [BN][32], rows of 64 B
// whose four 16-byte chunks are swizzled by (row >> 2) & 3; two stages (48 KB at most): three workgroups share a CU, so
// one's stores overlap another's loads - these layers are bound by HBM, not by the 3x MFMA count.
// DUAL (conv_igemm.hip's two-source idea, for the paired head's first block): the K dimension comes from TWO pixel-aligned
// pair tensors of equal width - K-steps [0, Cin / 32) from x (t2: conv3's input), the rest from x2 (the block input: the
// stride-1 downsample branch, resnet.py:134-141), weights concatenated along K, biases summed: relu([W3 | Wds] . [t2 ; x]
// + b3 + bds) in one launch, and the 4P-wide downsample tensor (a pair: 2 x 1 GB at batch 32) is never written or read.
template <int BM, int BN, int WGM, int WGN, bool XP, bool DUAL = false>
__global__ void __launch_bounds__(64 * WGM * WGN) conv_pair_kernel(const ConvArgs a) {
    typedef FP16 DT;
    typedef DT::frag_t frag_t;
    constexpr int BK = 32, RB = 64, CPR = 4, KS = 2;
    constexpr int NT = 64 * WGM * WGN;
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int NA = BM * CPR / NT, NB = BN * CPR / NT;
    constexpr int XPL = XP ? 2 : 1;
    constexpr int XS = BM * RB, WS = BN * RB;
    constexpr int WOFF = XPL * XS;
    constexpr int STAGE_BYTES = XPL * XS + 2 * WS;
    constexpr int EROW = TN * 128 + 16;
    static_assert(TM >= 1 && TN >= 1 && NA >= 1 && NB >= 1, "tile split");
    static_assert((BM * CPR) % NT == 0 && (BN * CPR) % NT == 0 && (NT / CPR) % 16 == 0, "chunk split");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WGN, wm = wave / WGN;
    const int lrow = lane & 31, lhi = lane >> 5;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n;   // n fastest: the channel tiles of one pixel tile share an XCD's L2
    const int tile_m = wg / a.tiles_n;

    const __amdgpu_buffer_rsrc_t rsrc_xh = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_xl =
        __builtin_amdgcn_make_buffer_rsrc((void*)(XP ? a.x_lo : a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_lo, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x2h = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? a.x2 : a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x2l =
        __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL && XP ? a.x2_lo : a.x), 0, a.x_bytes, 0x00020000);

    // per-lane source offsets: chunk slot tid % 4 of LDS row tid / 4 (+ i * NT/4) holds source chunk slot ^ swz(row)
    const int srcchunk = (tid & 3) ^ ((tid >> 4) & 3);
    const bool one_tap = (a.R * a.S == 1);
    int xbase[NA];
    uint32_t xmask[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = i * (NT / CPR) + tid / CPR;
        const int m = tile_m * BM + row;
        const bool mvalid = m < a.M;
        if (a.flat) {
            xbase[i] = (m * a.Cin + srcchunk * 8) * 2;
            xmask[i] = mvalid ? 1u : 0u;
        } else {
            const uint32_t mm = mvalid ? (uint32_t)m : 0u;
            const uint32_t b = fast_div_p(mm, a.div_ohw_mul, a.div_ohw_shr);
            const uint32_t rem = mm - b * (uint32_t)(a.OH * a.OW);
            const uint32_t oh = fast_div_p(rem, a.div_ow_mul, a.div_ow_shr);
            const uint32_t ow = rem - oh * (uint32_t)a.OW;
            const int ih0 = (int)oh * a.stride - a.pad;
            const int iw0 = (int)ow * a.stride - a.pad;
            xbase[i] = (((int)b * a.H + ih0) * a.W + iw0) * a.Cin * 2 + srcchunk * 16;
            auto range_bits = [](int lo, int hi) -> uint32_t {
                return hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
            };
            const uint32_t rbits = range_bits(max(0, -ih0), min(a.R, a.H - ih0));
            const uint32_t cbits = range_bits(max(0, -iw0), min(a.S, a.W - iw0));
            uint32_t mask = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) mask |= ((rbits >> r) & 1u) ? (cbits << (r * a.S)) : 0u;
            xmask[i] = mvalid ? mask : 0u;
        }
    }
    uint32_t wvoff[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = i * (NT / CPR) + tid / CPR;
        wvoff[i] = (uint32_t)(((tile_n * BN + row) * a.Ktot + srcchunk * 8) * 2);
    }

    auto issue = [&](int wstep, int tap, int koff, bool second, char* stage) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            char* dst = stage + (i * NT + wave * 64) * 16;
            const uint32_t v = ((xmask[i] >> tap) & 1u) ? (uint32_t)(xbase[i] + (one_tap ? 0 : koff)) : kOOBp;
            const int so = one_tap ? koff : 0;   // 1x1: the K-step rides in the scalar offset
            if (DUAL && second) {                // same pixel, same width: only the tensor differs
                dma16p(rsrc_x2h, dst, v, so);
                if (XP) dma16p(rsrc_x2l, dst + XS, v, so);
            } else {
                dma16p(rsrc_xh, dst, v, so);
                if (XP) dma16p(rsrc_xl, dst + XS, v, so);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            char* dst = stage + WOFF + (i * NT + wave * 64) * 16;
            dma16p(rsrc_wh, dst, wvoff[i], wstep * RB);
            dma16p(rsrc_wl, dst + WS, wvoff[i], wstep * RB);
        }
    };

    const int lswz = (lane >> 2) & 3;
    int loff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) loff[ks] = lrow * RB + (((2 * ks + lhi) ^ lswz) << 4);
    const int xfrag = (wm * TM * 32) * RB;
    const int wfrag = WOFF + (wn * TN * 32) * RB;

    f32x16_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_t b4 = *(const DIR_GLOBAL f32x4_t*)(a.bias + tile_n * BN + wn * TN * 32 + i * 32 + 8 * g + 4 * lhi);
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = b4[e];
        }

    auto compute = [&](const char* stage) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            frag_t wh[TN], wl[TN], xh[TM], xl[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                wh[i] = *(const frag_t*)(stage + wfrag + i * 32 * RB + loff[ks]);
                wl[i] = *(const frag_t*)(stage + wfrag + WS + i * 32 * RB + loff[ks]);
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                xh[j] = *(const frag_t*)(stage + xfrag + j * 32 * RB + loff[ks]);
                if (XP) xl[j] = *(const frag_t*)(stage + xfrag + XS + j * 32 * RB + loff[ks]);
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    acc[i][j] = DT::mfma32(wh[i], xh[j], acc[i][j]);
                    if (XP) acc[i][j] = DT::mfma32(wh[i], xl[j], acc[i][j]);
                    acc[i][j] = DT::mfma32(wl[i], xh[j], acc[i][j]);
                }
        }
    };

    // ---- residual tile (both planes) fetched up front: its HBM latency hides under the K loop (conv_igemm.hip's
    //      PRE_RES).  Read in the epilogue it cost eight exposed round trips per wave and tile: conv3 of the paired
    //      layer1 ran at 0.81 ms for 2.7 GB, 3.3 TB/s. -----------------------------------------------------------------
    constexpr int LPR = TN * 4, RPP = 64 / LPR, NPASS = 32 / RPP;
    const int ecol = (lane % LPR) * 8;
    const int erow = lane / LPR;
    const int n_glob = tile_n * BN + wn * TN * 32 + ecol;
    const int m_epi = tile_m * BM + wm * TM * 32;
    u32x4_t rres_h[TM][NPASS], rres_l[TM][NPASS];
    if (a.res) {
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int m = m_epi + j * 32 + pass * RPP + erow;
                const size_t o = (size_t)(m < a.M ? m : 0) * a.Cout + n_glob;   // clamped rows are never stored
                rres_h[j][pass] = gload16(a.res + o);
                if (a.res_lo) rres_l[j][pass] = gload16(a.res_lo + o);
            }
    }

    // ---- K loop: channel slice outermost, filter taps innermost (conv_igemm.hip's order); two slots -----------------
    const int cpb = a.Cin / BK;
    const int T = a.T;
    int tap = 0, cc = 0, r = 0, s = 0;
    // (DUAL: flat 1x1 only - tap == 0, cc counts K-steps over both sources, the second one starts at cc == cpb)
    auto second_now = [&]() { return DUAL && cc >= cpb; };
    auto koff_now = [&]() { return ((r * a.W + s) * a.Cin + (second_now() ? cc - cpb : cc) * BK) * 2; };
    auto wstep_now = [&]() { return tap * cpb + cc; };
    auto advance = [&]() {
        ++tap;
        if (++s == a.S) {
            s = 0;
            if (++r == a.R) {
                r = 0;
                tap = 0;
                ++cc;
            }
        }
    };
    issue(wstep_now(), tap, koff_now(), second_now(), smem);
    advance();
    for (int t = 0; t < T; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_barrier();   // stage t landed everywhere; everyone is done reading the other slot
        if (t + 1 < T) {
            issue(wstep_now(), tap, koff_now(), second_now(), smem + ((t + 1) & 1) * STAGE_BYTES);
            advance();
        }
        compute(smem + (t & 1) * STAGE_BYTES);
    }
    __syncthreads();   // the ring becomes epilogue staging
    Ovf<DT> ovf;

    // ---- epilogue: acc -> LDS (fp32, pixel-major) -> + residual pair -> ReLU -> split into (hi, lo) -> 16-byte stores
    char* ebase = smem + wave * (32 * EROW);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4_t v = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *(f32x4_t*)(ebase + lrow * EROW + (i * 32 + 8 * g + 4 * lhi) * 4) = v;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int mrow = pass * RPP + erow;
            const f32x4_t f0 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4);
            const f32x4_t f1 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4 + 16);
            const int m = m_epi + j * 32 + mrow;
            if (m < a.M) {
                float v[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
                const size_t o = (size_t)m * a.Cout + n_glob;
                if (a.res) {
                    const u32x4_t rh = rres_h[j][pass];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo, hi;
                        DT::unpack(rh[e], lo, hi);
                        v[2 * e] += lo;
                        v[2 * e + 1] += hi;
                    }
                    if (a.res_lo) {
                        const u32x4_t rl = rres_l[j][pass];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float lo, hi;
                            DT::unpack(rl[e], lo, hi);
                            v[2 * e] += lo;
                            v[2 * e + 1] += hi;
                        }
                    }
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                u32x4_t oh, ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t h, l;
                    split2(v[2 * e], v[2 * e + 1], h, l);
                    oh[e] = h;
                    ol[e] = l;
                }
                gstore16(a.y + o, oh);
                if (a.y_lo) gstore16(a.y_lo + o, ol);
                ovf.see(oh);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    ovf.flush(a.ovf);
}

// ---- 3x3 stride 1, 64 -> 64 channels, from an LDS-resident patch PAIR (layer1's conv2) --------------------------------------
// The implicit-GEMM form above re-fetches the activation tile once per filter tap and plane: 18 x the input through the
// CU's memory pipe (0.60 ms per launch at batch 32, 1.8 TB/s).  Here (conv_patch.hip's idea) a workgroup loads the
// (4 + 2) x (32 + 2) pixel patch of its 4 x 32 output tile ONCE per plane (hi, lo: 2 x 28 KB) and the nine taps read it at
// shifted pixel offsets; only the weights stream: one [64][64] slice per (tap, plane) step through a 3-slot ring of 8 KB.
// A hi step multiplies w_hi by both patch planes, the lo step that follows it w_lo by the hi plane: the same three
// products per term.  80 KB of LDS and 4 waves: two workgroups per CU, one's patch load under the other's MFMAs.
__global__ void __launch_bounds__(256, 2) conv_pair_patch64_kernel(const ConvArgs a) {
    typedef FP16 DT;
    typedef DT::frag_t frag_t;
    constexpr int TH = 4, TW = 32, PH = TH + 2, PW = TW + 2, PP = PH * PW;   // 204 patch pixels
    constexpr int C = 64, NTH = 256;
    constexpr int NPL = (PP * 8 + NTH - 1) / NTH;       // 7 DMA instructions per lane per plane
    constexpr int PLANE_BYTES = NPL * NTH * 16;         // 28672
    constexpr int WSTAGE = C * 128, NBW = C * 8 / NTH, NSTW = 3;
    constexpr int WOFF = 2 * PLANE_BYTES;
    constexpr int TN = 2;
    constexpr int EROW = TN * 128 + 16;
    constexpr int NS = 18;                              // (tap, plane) steps
    static_assert(WOFF + NSTW * WSTAGE <= 80 * 1024 && 4 * 32 * EROW <= WOFF, "two workgroups per CU; staging aliases the patch");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = output row of the tile
    const int lrow = lane & 31, lhi = lane >> 5;

    const int tiles_x = (a.OW + TW - 1) / TW;
    const int tiles_y = (a.OH + TH - 1) / TH;
    int wg = blockIdx.x;
    const int tx = wg % tiles_x;
    wg /= tiles_x;
    const int ty = wg % tiles_y;
    const int b = wg / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;

    const __amdgpu_buffer_rsrc_t rsrc_xh = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_xl = __builtin_amdgcn_make_buffer_rsrc((void*)a.x_lo, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_lo, 0, a.w_bytes, 0x00020000);

    // ---- patch pair: PP pixels x 128 B per plane, chunks XOR-swizzled with (p >> 1) & 7, loaded once ----------------------
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int P = i * NTH + tid;
        const int p = P >> 3, slot = P & 7;
        const int py = p / PW, px = p - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool ok = p < PP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const uint32_t v = ok ? (uint32_t)((((b * a.H + iy) * a.W + ix) * C + ((slot ^ ((p >> 1) & 7)) << 3)) * 2) : kOOBp;
        dma16p(rsrc_xh, smem + (i * NTH + wave * 64) * 16, v, 0);
        dma16p(rsrc_xl, smem + PLANE_BYTES + (i * NTH + wave * 64) * 16, v, 0);
    }
    // ---- weights: step sigma = (tap, plane) -> the [64][64] slice of that tap from w_hi / w_lo -----------------------------
    const int srcchunk = (tid & 7) ^ ((tid >> 4) & 7);
    uint32_t wvoff[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) wvoff[i] = (uint32_t)(((i * (NTH / 8) + (tid >> 3)) * a.Ktot + srcchunk * 8) * 2);
    auto issue_w = [&](int sigma, int slot) {
        const int tap = sigma >> 1;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            char* dst = smem + WOFF + slot * WSTAGE + (i * NTH + wave * 64) * 16;
            if (sigma & 1) {
                dma16p(rsrc_wl, dst, wvoff[i], tap * 128);
            } else {
                dma16p(rsrc_wh, dst, wvoff[i], tap * 128);
            }
        }
    };

    f32x16_t acc[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_t b4 = *(const DIR_GLOBAL f32x4_t*)(a.bias + i * 32 + 8 * g + 4 * lhi);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][4 * g + e] = b4[e];
        }
    const int wswz = (lane >> 1) & 7;
    int woffk[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) woffk[ks] = lrow * 128 + (((2 * ks + lhi) ^ wswz) << 4);

    issue_w(0, 0);
    issue_w(1, 1);
    int slot_c = 0;
    // One iteration per tap = a hi step and a lo step (a hand-off barrier each).  The patch fragments of a tap (xh, xl: the
    // patch is resident, they do not depend on the weight ring) are read ONCE for both steps, and those of the NEXT tap are
    // requested before the lo step's MFMAs so that their LDS latency hides under those.
    auto patch_frags = [&](int tap, frag_t* xh, frag_t* xl) {
        const int r = tap / 3, sx = tap - 3 * r;
        const int p = (wave + r) * PW + sx + lrow;          // patch pixel this lane reads
        const int swz = (p >> 1) & 7;
        const char* rowh = smem + p * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            xh[ks] = *(const frag_t*)(rowh + (((2 * ks + lhi) ^ swz) << 4));
            xl[ks] = *(const frag_t*)(rowh + PLANE_BYTES + (((2 * ks + lhi) ^ swz) << 4));
        }
    };
    auto hand_off = [&](int sigma) {
        if (sigma + 1 < NS) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NBW) : "memory");   // the newest stage may stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ring_barrier();   // the patch pair (issued first) and weight stage sigma have landed; everyone left stage sigma - 1
        if (sigma + 2 < NS) {
            int slot_n = slot_c + 2;
            if (slot_n >= NSTW) slot_n -= NSTW;
            issue_w(sigma + 2, slot_n);
        }
    };
    frag_t xh[2][4], xl[2][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int cur = tap & 1;
        // ---- hi step: w_hi . (x_hi + x_lo) ---------------------------------------------------------------------------------
        hand_off(2 * tap);
        if (tap == 0) patch_frags(0, xh[0], xl[0]);          // (the patch has only just landed)
        {
            const char* wst = smem + WOFF + slot_c * WSTAGE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    const frag_t wf = *(const frag_t*)(wst + i * 4096 + woffk[ks]);
                    acc[i] = DT::mfma32(wf, xh[cur][ks], acc[i]);
                    acc[i] = DT::mfma32(wf, xl[cur][ks], acc[i]);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this stage's LDS reads retired before the next barrier
        if (++slot_c == NSTW) slot_c = 0;
        // ---- lo step: w_lo . x_hi ------------------------------------------------------------------------------------------------
        hand_off(2 * tap + 1);
        {
            const char* wst = smem + WOFF + slot_c * WSTAGE;
            frag_t wl[4][TN];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < TN; ++i) wl[ks][i] = *(const frag_t*)(wst + i * 4096 + woffk[ks]);
            if (tap + 1 < 9) patch_frags(tap + 1, xh[cur ^ 1], xl[cur ^ 1]);   // next tap's pixels: 8 reads, newest in the queue
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < TN; ++i) acc[i] = DT::mfma32(wl[ks][i], xh[cur][ks], acc[i]);
        }
        // every LDS read retired before the next barrier (the next tap's patch reads had the eight MFMAs above to land; a
        // counted wait that leaves them in flight would depend on the order hipcc emits the reads in)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (++slot_c == NSTW) slot_c = 0;
    }
    ring_barrier();   // (every LDS read has retired, every DMA has landed: see the last step) the patch becomes epilogue staging
    Ovf<DT> ovf;

    char* ebase = smem + wave * (32 * EROW);
    constexpr int LPR = TN * 4, RPP = 64 / LPR, NPASS = 32 / RPP;
    const int ecol = (lane % LPR) * 8, erow = lane / LPR;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4_t v = {acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
            *(f32x4_t*)(ebase + lrow * EROW + (i * 32 + 8 * g + 4 * lhi) * 4) = v;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int oy = oy0 + wave;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int mrow = pass * RPP + erow;
        const f32x4_t f0 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4);
        const f32x4_t f1 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4 + 16);
        const int ox = ox0 + mrow;
        if (oy < a.OH && ox < a.OW) {
            float v[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            u32x4_t oh, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t h, l;
                split2(v[2 * e], v[2 * e + 1], h, l);
                oh[e] = h;
                ol[e] = l;
            }
            const size_t o = ((size_t)(b * a.OH + oy) * a.OW + ox) * C + ecol;
            gstore16(a.y + o, oh);
            if (a.y_lo) gstore16(a.y_lo + o, ol);
            ovf.see(oh);
        }
    }
    ovf.flush(a.ovf);
}

static bool pair_patch64_admissible(const ConvArgs& a) {
    return a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.H == a.OH && a.W == a.OW && a.Cin == 64 &&
           a.Cout == 64 && a.x_lo && !a.res && !a.x2;
}

static hipError_t launch_pair_patch64(const ConvArgs& a, hipStream_t stream) {
    constexpr int LDS = 2 * 7 * 256 * 16 + 3 * 64 * 128;   // patch pair + three weight slots = 80 KiB
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)conv_pair_patch64_kernel, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    const long blocks = (long)a.B * ((a.OH + 3) / 4) * ((a.OW + 31) / 32);
    hipLaunchKernelGGL(conv_pair_patch64_kernel, dim3((unsigned)blocks), dim3(256), LDS, stream, b);
    return hipGetLastError();
}

static void fastdiv_init_p(uint32_t d, uint32_t& mul, uint32_t& shr) {
    if (d <= 1) {
        mul = 0;
        shr = 0;
        return;
    }
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    mul = (uint32_t)(((1ull << (31 + l)) + d - 1) / d);
    shr = l - 1;
}

template <int BM, int BN, int WGM, int WGN, bool XP, bool DUAL = false>
static hipError_t launch_pair(const ConvArgs& a, hipStream_t stream) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int TN = BN / WGN / 32;
    constexpr int EROW = TN * 128 + 16;
    constexpr int STAGE_BYTES = ((XP ? 2 : 1) * BM + 2 * BN) * 64;
    constexpr int EPI_BYTES = (NT / 64) * 32 * EROW;
    constexpr int LDS = (2 * STAGE_BYTES > EPI_BYTES) ? 2 * STAGE_BYTES : EPI_BYTES;
    static_assert(LDS <= 64 * 1024, "two or three workgroups per CU");
    auto kern = conv_pair_kernel<BM, BN, WGM, WGN, XP, DUAL>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.T = a.Ktot / 32;
    b.tiles_m = ceil_div(a.M, BM);
    b.tiles_n = a.Cout / BN;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    b.flat = (a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.OH && a.W == a.OW);
    fastdiv_init_p((uint32_t)(a.OH * a.OW), b.div_ohw_mul, b.div_ohw_shr);
    fastdiv_init_p((uint32_t)a.OW, b.div_ow_mul, b.div_ow_shr);
    const int used = (b.T < 2 ? b.T : 2) * STAGE_BYTES;
    const int lds = used > EPI_BYTES ? used : EPI_BYTES;
    hipLaunchKernelGGL(kern, dim3(b.tiles_m * b.tiles_n), dim3(NT), lds, stream, b);
    return hipGetLastError();
}

// 128 pixels x 128 channels for the wide outputs (the pixel tile is fetched once per 128 channels), x 64 otherwise
// (64-channel tiles for the wide outputs too - three workgroups per CU instead of two - measured 4-6 % slower on every
// 1x1 of the paired layer1, gpurun_out/r4h)
static bool pair_wide(const ConvArgs& a) { return a.Cout % 128 == 0; }

// DIRTORCH_AMD_NO_PAIR_PATCH=1: layer1's 3x3 back on the implicit-GEMM form (A/B and bisecting; read per call)
static bool use_pair_patch64(const ConvArgs& a) {
    return pair_patch64_admissible(a) && !env().no_pair_patch;
}

const char* conv_pair_variant_name(const ConvArgs& a) {
    if (a.x2) return "128x128_xw/dual";
    if (use_pair_patch64(a)) return "128x64_patch3x3_xw";
    return pair_wide(a) ? (a.x_lo ? "128x128_xw" : "128x128_w") : (a.x_lo ? "128x64_xw" : "128x64_w");
}

int conv_pair_launch(const ConvArgs& a, hipStream_t stream) {
    if (!a.x || !a.w || !a.w_lo || !a.bias || !a.y) return fail(DIR_ERR_INVALID, "conv_pair: null pointer");
    if (a.Cin % 32 != 0 || a.Cout % 64 != 0) return fail(DIR_ERR_INVALID, "conv_pair: Cin % 32, Cout % 64 required");
    if (a.R < 1 || a.S < 1 || a.R > 4 || a.S > 4 || a.stride < 1 || a.pad < 0)
        return fail(DIR_ERR_INVALID, "conv_pair: bad filter geometry");
    if (a.res_lo && !a.res) return fail(DIR_ERR_INVALID, "conv_pair: res_lo without res");
    if ((long)a.B * a.H * a.W * a.Cin >= (1L << 30) || (long)a.M * a.Cout >= (1L << 30) ||
        (long)a.Cout * a.Ktot >= (1L << 30))
        return fail(DIR_ERR_INVALID, "conv_pair: tensor exceeds 2^31 bytes; lower the batch");
    const void* ptrs[] = {a.x, a.x_lo, a.w, a.w_lo, a.res, a.res_lo, a.y, a.y_lo, a.bias};
    for (const void* p : ptrs)
        if ((uintptr_t)p & 15) return fail(DIR_ERR_INVALID, "conv_pair: tensors must be 16-byte aligned");
    hipError_t e;
    if (use_pair_patch64(a)) {
        e = launch_pair_patch64(a, stream);
    } else if (a.x2) {   // two-source form: conv3 + the stride-1 downsample of the paired head's first block
        const bool flat = a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.OH && a.W == a.OW;
        if (!flat || a.Cin2 != a.Cin || a.Ktot != 2 * a.Cin || !a.x_lo || !a.x2_lo || a.res || a.Cout % 128 != 0 ||
            ((uintptr_t)a.x2 & 15) || ((uintptr_t)a.x2_lo & 15))
            return fail(DIR_ERR_INVALID, "conv_pair: the two-source form takes two pixel-aligned pair tensors of equal width, "
                                         "1x1 stride 1, Cout % 128 == 0, no residual");
        e = launch_pair<128, 128, 2, 2, true, true>(a, stream);
    } else if (pair_wide(a))
        e = a.x_lo ? launch_pair<128, 128, 2, 2, true>(a, stream) : launch_pair<128, 128, 2, 2, false>(a, stream);
    else
        e = a.x_lo ? launch_pair<128, 64, 2, 2, true>(a, stream) : launch_pair<128, 64, 2, 2, false>(a, stream);
    if (e != hipSuccess) return fail(DIR_ERR_HIP, std::string("conv_pair launch: ") + hipGetErrorString(e));
    return DIR_OK;
}

// ---- stem: 7x7 s2 conv + BN + ReLU + 3x3 s2 max-pool on pairs ---------------------------------------------------------
// stem_pool.hip's one-tile-per-workgroup form (a 3 x 15 tile of pooled pixels from an 11 x 35 patch of the 2x2
// space-to-depth image, filter row R = one K-step of 64) with every operand a pair: patch hi / lo (2 x 16 KB) and
// filter hi / lo (2 x 32 KB) in LDS, three MFMAs per term.  The 8 x 32 conv tile is kept in FP32 (64 KB over the dead
// patch + filter), pooled in fp32, and only the pooled pixels are split into (hi, lo) and written.
struct StemPairArgs {
    const uint16_t *xh, *xl;   // s2d image pair [B, H2, W2, 16]
    const uint16_t *wh, *wl;   // [64][4][4][16]
    const float* bias;
    uint16_t *yh, *yl;         // pooled pair [B, PH, PW, 64]
    int B, H2, W2, OH, OW, PH, PW;
    uint32_t x_bytes;
    int* ovf;
};

__global__ void __launch_bounds__(256) stem_pool_pair_kernel(const StemPairArgs a) {
    typedef FP16 DT;
    typedef DT::frag_t frag_t;
    constexpr int PTH = 3, PTW = 15;
    constexpr int TH = 8, TW = 32;
    constexpr int QW = TW + 3;
    constexpr int QP = (TH + 3) * QW;
    constexpr int PLANE = 512 * 16;            // one channel-half plane of the patch
    constexpr int PATCH = 2 * PLANE;           // 16 KiB per patch (hi or lo)
    constexpr int WOFF = 2 * PATCH;            // filter hi at 32 KiB, filter lo at 64 KiB
    constexpr int WBYTES = 4 * 8192;
    static_assert(TH * TW * 256 <= WOFF + WBYTES, "the fp32 conv tile aliases the patches + the hi filter");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;

    const int tiles_x = (a.PW + PTW - 1) / PTW;
    const int tiles_y = (a.PH + PTH - 1) / PTH;
    int wg = blockIdx.x;
    const int tx = wg % tiles_x;
    wg /= tiles_x;
    const int ty = wg % tiles_y;
    const int b = wg / tiles_y;
    const int ph0 = ty * PTH, pw0 = tx * PTW;
    const int oy0 = 2 * ph0 - 1, ox0 = 2 * pw0 - 1;

    const __amdgpu_buffer_rsrc_t rsrc_xh = __builtin_amdgcn_make_buffer_rsrc((void*)a.xh, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_xl = __builtin_amdgcn_make_buffer_rsrc((void*)a.xl, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.wh, 0, 64 * 256 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.wl, 0, 64 * 256 * 2, 0x00020000);

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int P = i * 256 + tid;
        const int plane = P >> 9, p = P & 511;
        const int py = p / QW, px = p - py * QW;
        const int iy = oy0 - 2 + py, ix = ox0 - 2 + px;
        const bool ok = p < QP && (unsigned)iy < (unsigned)a.H2 && (unsigned)ix < (unsigned)a.W2;
        const uint32_t v = ok ? (uint32_t)((((b * a.H2 + iy) * a.W2 + ix) * 16 + plane * 8) * 2) : kOOBp;
        dma16p(rsrc_xh, smem + (i * 256 + wave * 64) * 16, v, 0);
        dma16p(rsrc_xl, smem + PATCH + (i * 256 + wave * 64) * 16, v, 0);
    }
    const int srcchunk = (tid & 7) ^ ((tid >> 4) & 7);
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = i * 32 + (tid >> 3);
            const uint32_t v = (uint32_t)((n * 256 + srcchunk * 8) * 2);
            dma16p(rsrc_wh, smem + WOFF + R * 8192 + (i * 256 + wave * 64) * 16, v, R * 128);
            dma16p(rsrc_wl, smem + WOFF + WBYTES + R * 8192 + (i * 256 + wave * 64) * 16, v, R * 128);
        }

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
            frag_t wh[2], wl[2], xh[2], xl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int o = WOFF + R * 8192 + (i * 32 + lrow) * 128 + (((2 * ks + lhi) ^ wswz) << 4);
                wh[i] = *(const frag_t*)(smem + o);
                wl[i] = *(const frag_t*)(smem + o + WBYTES);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = (wave * 2 + j + R) * QW + lrow + ks;   // patch pixel (oy + R, ox + ks)
                xh[j] = *(const frag_t*)(smem + lhi * PLANE + p * 16);
                xl[j] = *(const frag_t*)(smem + PATCH + lhi * PLANE + p * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = DT::mfma32(wh[i], xh[j], acc[i][j]);
                    acc[i][j] = DT::mfma32(wh[i], xl[j], acc[i][j]);
                    acc[i][j] = DT::mfma32(wl[i], xh[j], acc[i][j]);
                }
        }
    }
    __syncthreads();   // patches and filters are dead: the fp32 conv tile takes their place

    // ReLU (signed-integer max on the bit pattern: negatives and -0 -> +0), out-of-image conv outputs -> 0 (exact for the
    // max: every pool window holds a real post-ReLU value); pixel-major rows of 256 B, 16-byte chunk index ^ (x & 7)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tyy = wave * 2 + j;
        const int oy = oy0 + tyy, ox = ox0 + lrow;
        const int inmask = ((unsigned)oy < (unsigned)a.OH && (unsigned)ox < (unsigned)a.OW) ? -1 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[i][j][4 * g + e];
                    v[e] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, t), 0) & inmask);
                }
                const int c16 = 8 * i + 2 * g + lhi;   // 16-byte chunk (4 channels) of this pixel's 256-byte row
                *(f32x4_t*)(smem + (tyy * TW + lrow) * 256 + ((c16 ^ (lrow & 7)) << 4)) = v;
            }
    }
    __syncthreads();

    Ovf<DT> ovf;
    for (int it = tid; it < PTH * PTW * 8; it += 256) {
        const int c8 = it & 7;
        const int pp = it >> 3;
        const int py = pp / PTW, px = pp - py * PTW;
        const int ph = ph0 + py, pw = pw0 + px;
        if (ph >= a.PH || pw >= a.PW) continue;
        f32x4_t m0 = {0.f, 0.f, 0.f, 0.f}, m1 = {0.f, 0.f, 0.f, 0.f};   // post-ReLU values are >= 0
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int yy = 2 * py + dy, xx = 2 * px + dx;
                const char* row = smem + (yy * TW + xx) * 256;
                const f32x4_t u0 = *(const f32x4_t*)(row + (((2 * c8) ^ (xx & 7)) << 4));
                const f32x4_t u1 = *(const f32x4_t*)(row + (((2 * c8 + 1) ^ (xx & 7)) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m0[e] = fmaxf(m0[e], u0[e]);
                    m1[e] = fmaxf(m1[e], u1[e]);
                }
            }
        u32x4_t oh, ol;
        const float mv[8] = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t h, l;
            split2(mv[2 * e], mv[2 * e + 1], h, l);
            oh[e] = h;
            ol[e] = l;
        }
        const size_t o = ((size_t)(b * a.PH + ph) * a.PW + pw) * 64 + c8 * 8;
        gstore16(a.yh + o, oh);
        gstore16(a.yl + o, ol);
        ovf.see(oh);
    }
    ovf.flush(a.ovf);
}

// ---- persistent form --------------------------------------------------------------------------------------------------
// The one-tile-per-workgroup kernel above re-loads 64 KB of filter per 3 x 15 pooled tile and runs load / MFMA / pool
// strictly one after the other (1.42 ms at batch 32).  Here one 8-wave workgroup per CU walks a strided list of tiles with
// the FILTER PAIR IN REGISTERS: wave w owns channel tile w & 1 and conv rows 2 (w >> 1), 2 (w >> 1) + 1 and holds that
// channel tile's 16 hi + 16 lo fragments (128 VGPRs, fetched once, straight in MFMA operand layout).  Two patch-pair
// buffers (the next tile's patch is in flight while this one is multiplied and pooled) + the fp32 conv tile: 128 KiB.
__global__ void __launch_bounds__(512) stem_pool_pair_persist_kernel(const StemPairArgs a) {
    typedef FP16 DT;
    typedef DT::frag_t frag_t;
    constexpr int PTH = 3, PTW = 15;
    constexpr int TH = 8, TW = 32;
    constexpr int QW = TW + 3;
    constexpr int QP = (TH + 3) * QW;
    constexpr int PLANE = 512 * 16;
    constexpr int PATCH = 2 * PLANE;            // one plane pair of channel halves: 16 KiB (hi or lo)
    constexpr int PBUF = 2 * PATCH;             // hi + lo
    constexpr int TILE_OFF = 2 * PBUF;          // conv tile behind the two patch buffers
    constexpr int BIAS_OFF = TILE_OFF + TH * TW * 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int ci = wave & 1, rp = wave >> 1;

    const int tiles_x = (a.PW + PTW - 1) / PTW;
    const int tiles_y = (a.PH + PTH - 1) / PTH;
    const int ntiles = a.B * tiles_y * tiles_x;

    const __amdgpu_buffer_rsrc_t rsrc_xh = __builtin_amdgcn_make_buffer_rsrc((void*)a.xh, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_xl = __builtin_amdgcn_make_buffer_rsrc((void*)a.xl, 0, a.x_bytes, 0x00020000);

    frag_t wfh[4][4], wfl[4][4];
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const size_t o = (size_t)(ci * 32 + lrow) * 256 + R * 64 + ks * 16 + lhi * 8;
            wfh[R][ks] = __builtin_bit_cast(frag_t, gload16(a.wh + o));
            wfl[R][ks] = __builtin_bit_cast(frag_t, gload16(a.wl + o));
        }
    float* lbias = (float*)(smem + BIAS_OFF);
    if (tid < 64) lbias[tid] = a.bias[tid];

    auto issue_patch = [&](int tile, char* dst) {
        int wg = tile;
        const int tx = wg % tiles_x;
        wg /= tiles_x;
        const int ty = wg % tiles_y;
        const int b = wg / tiles_y;
        const int oy0 = 2 * ty * PTH - 1, ox0 = 2 * tx * PTW - 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int P = i * 512 + tid;
            const int plane = P >> 9, p = P & 511;
            const int py = p / QW, px = p - py * QW;
            const int iy = oy0 - 2 + py, ix = ox0 - 2 + px;
            const bool ok = p < QP && (unsigned)iy < (unsigned)a.H2 && (unsigned)ix < (unsigned)a.W2;
            const uint32_t v = ok ? (uint32_t)((((b * a.H2 + iy) * a.W2 + ix) * 16 + plane * 8) * 2) : kOOBp;
            if (DIR_STEMP_ABL & 4) continue;   // (experiment: no patch DMA)
            dma16p(rsrc_xh, dst + (i * 512 + wave * 64) * 16, v, 0);
            dma16p(rsrc_xl, dst + PATCH + (i * 512 + wave * 64) * 16, v, 0);
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    Ovf<DT> ovf;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // filter in registers, bias in LDS, before the counted waits
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {   // pin: the waits for these loads must not be re-executed inside the loop
            asm volatile("" : "+v"(wfh[R][ks]));
            asm volatile("" : "+v"(wfl[R][ks]));
        }
    issue_patch(tile, smem);
    int cur = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const char* pbuf = smem + cur * PBUF;
        // the other buffer was last read by the MFMA phase of the previous tile, which every wave left before the barrier
        // in front of that tile's pooling
        if (next < ntiles) {
            issue_patch(next, smem + (cur ^ 1) * PBUF);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // this tile's patch; the 4 newest ops may fly
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ring_barrier();   // patch landed everywhere; everyone is done pooling the previous tile

        f32x16_t acc[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_t b4 = *(const f32x4_t*)(lbias + ci * 32 + 8 * g + 4 * lhi);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][4 * g + e] = b4[e];
        }
#pragma unroll
        for (int R = 0; R < 4; ++R)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                frag_t xh[2], xl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int p = (rp * 2 + j + R) * QW + lrow + ks;
                    xh[j] = *(const frag_t*)(pbuf + lhi * PLANE + p * 16);
                    xl[j] = *(const frag_t*)(pbuf + PATCH + lhi * PLANE + p * 16);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (DIR_STEMP_ABL & 1) continue;   // (experiment: no MFMAs, no fragment reads)
                    acc[j] = DT::mfma32(wfh[R][ks], xh[j], acc[j]);
                    acc[j] = DT::mfma32(wfh[R][ks], xl[j], acc[j]);
                    acc[j] = DT::mfma32(wfl[R][ks], xh[j], acc[j]);
                }
                // (issue order measured - weight fragment held over both rows with alternating accumulators, or the pixel
                // fragment held: 981-999 / 983-994 / 994-1009 us - no effect here: gpurun_out/r5stemorder)
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
            const int tyy = rp * 2 + j;
            const int oy = oy0 + tyy, ox = ox0 + lrow;
            const int inmask = ((unsigned)oy < (unsigned)a.OH && (unsigned)ox < (unsigned)a.OW) ? -1 : 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[j][4 * g + e];
                    v[e] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, t), 0) & inmask);
                }
                const int c16 = 8 * ci + 2 * g + lhi;
                *(f32x4_t*)(ctile + (tyy * TW + lrow) * 256 + ((c16 ^ (lrow & 7)) << 4)) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ring_barrier();   // conv tile complete; every wave is past its patch reads

        for (int it = tid; it < ((DIR_STEMP_ABL & 2) ? 0 : PTH * PTW * 8); it += 512) {   // (experiment bit 2: no pooling phase)
            const int c8 = it & 7;
            const int pp = it >> 3;
            const int py = pp / PTW, px = pp - py * PTW;
            const int ph = ph0 + py, pw = pw0 + px;
            if (ph >= a.PH || pw >= a.PW) continue;
            f32x4_t m0 = {0.f, 0.f, 0.f, 0.f}, m1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int yy = 2 * py + dy, xx = 2 * px + dx;
                    const char* row = ctile + (yy * TW + xx) * 256;
                    const f32x4_t u0 = *(const f32x4_t*)(row + (((2 * c8) ^ (xx & 7)) << 4));
                    const f32x4_t u1 = *(const f32x4_t*)(row + (((2 * c8 + 1) ^ (xx & 7)) << 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        m0[e] = fmaxf(m0[e], u0[e]);
                        m1[e] = fmaxf(m1[e], u1[e]);
                    }
                }
            u32x4_t oh, ol;
            const float mv[8] = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t hh, ll;
                split2(mv[2 * e], mv[2 * e + 1], hh, ll);
                oh[e] = hh;
                ol[e] = ll;
            }
            const size_t o = ((size_t)(b * a.PH + ph) * a.PW + pw) * 64 + c8 * 8;
            gstore16(a.yh + o, oh);
            gstore16(a.yl + o, ol);
            ovf.see(oh);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // pooling reads retired before the next tile's barrier
        cur ^= 1;
    }
    ovf.flush(a.ovf);
}

int stem_pool_pair_launch(const void* s2d_hi, const void* s2d_lo, const void* w_hi, const void* w_lo, const float* bias,
                          void* y_hi, void* y_lo, int B, int H2, int W2, int OH, int OW, hipStream_t stream, int* ovf) {
    if (!s2d_hi || !s2d_lo || !w_hi || !w_lo || !bias || !y_hi || !y_lo)
        return fail(DIR_ERR_INVALID, "stem_pool_pair: null pointer");
    if ((size_t)B * H2 * W2 * 32 >= (1ull << 31))
        return fail(DIR_ERR_INVALID, "stem_pool_pair: input exceeds 2^31 bytes; lower the batch");
    StemPairArgs a;
    a.xh = (const uint16_t*)s2d_hi;
    a.xl = (const uint16_t*)s2d_lo;
    a.wh = (const uint16_t*)w_hi;
    a.wl = (const uint16_t*)w_lo;
    a.bias = bias;
    a.yh = (uint16_t*)y_hi;
    a.yl = (uint16_t*)y_lo;
    a.B = B; a.H2 = H2; a.W2 = W2; a.OH = OH; a.OW = OW;
    a.PH = (OH - 1) / 2 + 1;
    a.PW = (OW - 1) / 2 + 1;
    a.x_bytes = (uint32_t)((size_t)B * H2 * W2 * 32);
    a.ovf = ovf;
    const long blocks = (long)B * ((a.PH + 2) / 3) * ((a.PW + 14) / 15);
    const bool v1 = env().stem_v1;   // A/B and bisecting (dir_reload_env after flipping it)
    // round 6, the default: stem_u8.hip's kernel structure on pairs (tiles walk down column strips, the max-pool runs in registers,
    // two 4-wave workgroups per CU; same sums in the same order - bit-identical); DIRTORCH_AMD_STEM_PAIR_OLD keeps the form below
    if (!v1 && !env().stem_pair_old)
        return stem_pool_pair_walk_launch(s2d_hi, s2d_lo, w_hi, w_lo, bias, y_hi, y_lo, B, H2, W2, OH, OW, stream, ovf);
    if (!v1) {
        constexpr int LDSP = 2 * 2 * 2 * 512 * 16 + 8 * 32 * 256 + 256;   // two patch pairs + the fp32 conv tile + bias
        static std::atomic<uint64_t> attr_p{0};
        DIR_HIP_CHECK(ensure_dynamic_lds((const void*)stem_pool_pair_persist_kernel, LDSP, attr_p));
        const long grid = blocks < cu_count() ? blocks : cu_count();
        hipLaunchKernelGGL(stem_pool_pair_persist_kernel, dim3((unsigned)grid), dim3(512), LDSP, stream, a);
        DIR_HIP_CHECK(hipGetLastError());
        return DIR_OK;
    }
    constexpr int LDS = 2 * 2 * 512 * 16 + 2 * 4 * 8192;   // patch pair + filter pair = 96 KiB
    static std::atomic<uint64_t> attr_done{0};
    DIR_HIP_CHECK(ensure_dynamic_lds((const void*)stem_pool_pair_kernel, LDS, attr_done));
    hipLaunchKernelGGL(stem_pool_pair_kernel, dim3((unsigned)blocks), dim3(256), LDS, stream, a);
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

// ---- image -> space-to-depth pair -------------------------------------------------------------------------------------
// prep_input (pointwise.hip) with both planes written: the normalised pixel (u8 / 255 - mean) / std - or the caller's
// fp32 value - is kept to ~22 bits instead of 11.
template <int FMT>
__global__ void prep_input_pair_kernel(const void* __restrict__ img, uint16_t* __restrict__ out_hi,
                                       uint16_t* __restrict__ out_lo, int B, int H, int W, int H2, int W2, float m0,
                                       float m1, float m2, float s0, float s1, float s2) {
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
                    } else {   // ToTensor (/255) then Normalize: dirtorch/utils/transforms.py:617-623
                        const uint8_t u = ((const uint8_t*)img)[(((size_t)b * H + y) * W + x) * 3 + c];
                        f = ((float)u / 255.f - mean[c]) / stdv[c];
                    }
                    v[(dy * 2 + dx) * 3 + c] = f;
                }
            }
        }
    }
    u32x4_t h0, h1, l0, l1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        uint32_t h, l;
        split2(v[2 * e], v[2 * e + 1], h, l);
        h0[e] = h;
        l0[e] = l;
        split2(v[8 + 2 * e], v[8 + 2 * e + 1], h, l);
        h1[e] = h;
        l1[e] = l;
    }
    gstore16(out_hi + idx * 16, h0);
    gstore16(out_hi + idx * 16 + 8, h1);
    gstore16(out_lo + idx * 16, l0);
    gstore16(out_lo + idx * 16 + 8, l1);
}

int prep_input_pair(const void* img, int fmt, const float* mean3, const float* std3, void* out_hi, void* out_lo, int B,
                    int H, int W, hipStream_t stream) {
    if (!img || !out_hi || !out_lo) return fail(DIR_ERR_INVALID, "prep_input_pair: null pointer");
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const long total = (long)B * H2 * W2;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    float m[3] = {0, 0, 0}, s[3] = {1, 1, 1};
    if (fmt == DIR_IMG_U8_NHWC) {
        if (!mean3 || !std3) return fail(DIR_ERR_INVALID, "prep_input_pair: u8 input needs mean/std");
        for (int c = 0; c < 3; ++c) {
            m[c] = mean3[c];
            s[c] = std3[c];
        }
        hipLaunchKernelGGL((prep_input_pair_kernel<DIR_IMG_U8_NHWC>), dim3(blocks), dim3(256), 0, stream, img,
                           (uint16_t*)out_hi, (uint16_t*)out_lo, B, H, W, H2, W2, m[0], m[1], m[2], s[0], s[1], s[2]);
    } else if (fmt == DIR_IMG_F32_NCHW) {
        hipLaunchKernelGGL((prep_input_pair_kernel<DIR_IMG_F32_NCHW>), dim3(blocks), dim3(256), 0, stream, img,
                           (uint16_t*)out_hi, (uint16_t*)out_lo, B, H, W, H2, W2, m[0], m[1], m[2], s[0], s[1], s[2]);
    } else {
        return fail(DIR_ERR_INVALID, "prep_input_pair: bad image format");
    }
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

}  // namespace dir
