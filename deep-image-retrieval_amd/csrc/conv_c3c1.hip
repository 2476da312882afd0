// conv_c3c1.hip — the seam between two bottlenecks in ONE kernel (gfx950):
//
//     out  = relu(conv3(t2) + bias3 + residual)          [M, 4P]   (closes block b,   resnet.py:78-85)
//     t1'  = relu(conv1'(out) + bias1)                    [M, P]    (opens  block b+1, resnet.py:70-72)
//
// Un-fused, the 4P-wide block output is written by conv3 and read back twice (conv1' and, later, the
// residual of block b+1).  In layer1 / layer2 both 1x1 convs are HBM-bound (AI 60-100 FLOP/B), and the
// read by conv1' is a quarter of everything the pair moves: t2 (P) + residual (4P) + out (4P) in,
// out (4P) + t1' (P) back = 14 P bytes per pixel un-fused, 10 P fused.  Here a persistent 8-wave
// workgroup keeps BOTH weight matrices in registers (W3: 4P x P, W1: P x 4P; 8 + 8 KB per wave at
// P = 128), streams 64-pixel tiles of t2, and hands the rounded output tile to conv1' through LDS:
//
//   phase A  (conv_wreg.hip's loop)  wave w owns output channels [w * P/2, (w+1) * P/2): MFMA over the
//            LDS t2 tile, fp32 staging per wave, + bias + residual (prefetched one tile ahead) + ReLU,
//            16-byte stores of `out` to HBM AND of the same 16 bytes into the LDS out-tile, laid out as
//            the next GEMM's B operand (64-channel blocks of 128-byte pixel rows, XOR-swizzled chunks)
//   phase B  the 8 waves split conv1' as (n-tile, K-half[, strip]); each multiplies its K half of the
//            out-tile by its register-resident slice of W1; the two K halves meet through the staging
//            area in fp32 (fixed order: results are run-to-run identical); + bias + ReLU, 16-byte stores
//
// DS form (first block of layer1): the residual of that block is itself a 1x1 convolution of the block
// input (downsample + BN, resnet.py:134-141).  Un-fused it writes a 4P-wide tensor that conv3 reads
// straight back; here it is simply 64 more K of the same GEMM - out = relu([W3 | Wds] . [t2 ; x] + (b3 +
// bds)) - so the block input (64 channels) is staged next to t2, the downsample weights sit in
// registers next to W3, and the 4P-wide residual tensor never exists (2 x 4P bytes per pixel less, one
// launch less; one 16-bit rounding less than the reference's storage points).
//
// WP3 / WP1 (DIR_FP16P, fp16 only): the weights of conv3 (+ downsample) / of conv1' are fp16 PAIRS (hi + lo planes, ~22 bits:
// csrc/conv_pair.hip); the lo planes sit in registers next to the hi ones and every product term costs a second MFMA - no
// bytes, and little time on these HBM-bound seams (MfmaUtil 0.13-0.24 single; measured at batch 32: 0.51-0.59 ms against
// 0.47-0.55, the DS form 0.52-0.58 against 0.38 - its MFMA phases sit between barriers).  In the DS form the block input (the stem's pooled output)
// is a pair too: its lo plane is staged as a third 64-channel K block that multiplies the downsample's HI weights (the
// lo x lo term, 2^-22 relative, is dropped as everywhere in the paired head).
//
// conv1' consumes exactly the 16-bit values stored to HBM, so the result equals the two-kernel path up
// to the fp32 summation order of the K halves.  Weights of layer3 (1 MB per seam) do not fit the 512 KB
// register file: that seam streams them from L2 through an LDS ring instead (conv_seam3.hip).
#include "dir_common.h"
#include "conv_igemm.h"

namespace dir {

static constexpr uint32_t kOOBf = 0x80000000u;

// P2 = planes of the block that conv1' opens: P inside a stage, 2 P across the layer1 -> layer2 boundary
// (layer2's first conv1 is 1x1 stride 1 over the 256-wide layer1 output; the stride sits in its conv2).
template <class DT, int P, bool DS, int P2, bool WP3 = false, bool WP1 = false>
__global__ void __launch_bounds__(512) conv_c3c1_kernel(const ConvArgs a) {
    static_assert(P == 64 || P == 128, "planes");
    static_assert(P2 == P || (P == 64 && P2 == 128), "conv1' width");
    static_assert(!DS || (P == 64 && P2 == 64), "the downsample form is layer1's: 64 + 64 input channels");
    static_assert((!WP3 && !WP1) || P == 64, "paired weights: layer1's seams");
    constexpr int KW = P + (DS ? 64 : 0);       // contraction length of the weight matrix (t2 channels [+ block input])
    constexpr int KA = KW + (DS && WP3 ? 64 : 0);   // ... of the staged tile: + the block input's lo plane
    constexpr int C4 = 4 * P;                   // block width
    constexpr int BM = 64, NT = 512;
    constexpr int CW = C4 / 8;                  // phase A: output channels per wave (32 / 64)
    constexpr int TA = CW / 32;                 // ... as 32-channel MFMA tiles (1 / 2)
    constexpr int KSA = KA / 16;                // phase A k-slices (4 / 8 / 12)
    constexpr int KSW = KW / 16;                // ... that have weights of their own
    constexpr int XBUF = BM * KA * 2;           // one t2 (+ x) tile: KA/64 blocks of [64 px][128 B]
    constexpr int NX = XBUF / 16 / NT;          // staging loads per lane per tile (1 / 2)
    constexpr int EROW = 32 * 4 + 16;           // staging row: 32 fp32 + pad
    constexpr int EPI_OFF = 2 * XBUF;
    constexpr int OUT_OFF = EPI_OFF + 8 * 32 * EROW;
    constexpr int OUTB = BM * C4 * 2;           // the out-tile: C4/64 blocks of [64 px][128 B]
    constexpr int BIAS_OFF = OUT_OFF + OUTB;    // bias3 [C4] then bias1 [P2], fp32
    constexpr int KSB = (C4 / 2) / 16;          // phase B k-slices per K half (8 / 16)
    typedef typename DT::frag_t frag_t;
    static_assert(NX >= 1 && XBUF % (16 * NT) == 0, "tile split");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x2 =      // DS: the block input [M][64]
        __builtin_amdgcn_make_buffer_rsrc((void*)(DS ? a.x2 : a.x), 0, DS ? (uint32_t)((size_t)a.M * 128) : a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x2l =     // DS + WP3: its lo plane
        __builtin_amdgcn_make_buffer_rsrc((void*)(DS && WP3 ? a.x2_lo : a.x), 0,
                                          DS && WP3 ? (uint32_t)((size_t)a.M * 128) : a.x_bytes, 0x00020000);
    const uint32_t y_bytes = (uint32_t)((size_t)a.M * C4 * 2);
    const uint32_t y2_bytes = (uint32_t)((size_t)a.M * P2 * 2);
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.y2, 0, y2_bytes, 0x00020000);

    const int mt = (a.M + BM - 1) / BM;
    const int per = gridDim.x;
    int tile = blockIdx.x;
    if (tile >= mt) return;
    Ovf<DT> ovf;
    const int n_wave = wave * CW;               // phase A: first output channel of this wave

    // ---- weights -> registers, once --------------------------------------------------------------
    frag_t w3[TA][KSW], w3l[WP3 ? TA : 1][WP3 ? KSW : 1];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
            w3[i][ks] = *(const DIR_GLOBAL frag_t*)(a.w + (size_t)(n_wave + i * 32 + lrow) * KW + ks * 16 + 8 * lhi);
            if (WP3)
                w3l[i][ks] = *(const DIR_GLOBAL frag_t*)(a.w_lo + (size_t)(n_wave + i * 32 + lrow) * KW + ks * 16 + 8 * lhi);
        }
    // phase B roles: P2 = 128: (n-tile = w & 3, K half = w >> 2), both 32-pixel strips;
    //                P2 =  64: (n-tile = w & 1, strip = (w >> 1) & 1, K half = w >> 2)
    constexpr bool WIDE = P2 == 128;
    const int nt = WIDE ? (wave & 3) : (wave & 1);
    const int kh = wave >> 2;
    const int jb = WIDE ? 0 : ((wave >> 1) & 1);
    frag_t w1[KSB], w1l[WP1 ? KSB : 1];
#pragma unroll
    for (int ks = 0; ks < KSB; ++ks) {
        w1[ks] = *(const DIR_GLOBAL frag_t*)(a.w2 + (size_t)(nt * 32 + lrow) * C4 + kh * (C4 / 2) + ks * 16 + 8 * lhi);
        if (WP1)
            w1l[ks] = *(const DIR_GLOBAL frag_t*)(a.w2_lo + (size_t)(nt * 32 + lrow) * C4 + kh * (C4 / 2) + ks * 16 + 8 * lhi);
    }
    // pin: the waits for these loads must not be re-executed inside the loop (conv_wreg.hip)
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
            asm volatile("" : "+v"(w3[i][ks]));
            if (WP3) asm volatile("" : "+v"(w3l[i][ks]));
        }
#pragma unroll
    for (int ks = 0; ks < KSB; ++ks) {
        asm volatile("" : "+v"(w1[ks]));
        if (WP1) asm volatile("" : "+v"(w1l[ks]));
    }

    // ---- per-lane constants -----------------------------------------------------------------------
    const int spix = (tid >> 3) & 63, sslot = tid & 7;                    // t2 staging: row, 16-byte chunk
    const int sdst = spix * 128 + ((sslot ^ ((spix >> 1) & 7)) << 4);
    const int lswz = (lane >> 1) & 7;
    const int lbase = lrow * 128;
    const int ecol = (lane & 3) * 8, erow = lane >> 2;                    // row layout: 4 lanes x 8 ch, 16 rows per pass
    float* const sbias3 = (float*)(smem + BIAS_OFF);
    float* const sbias1 = sbias3 + C4;
    for (int i = tid; i < C4; i += NT) sbias3[i] = a.bias[i];
    if (tid < P2) sbias1[tid] = a.bias2[tid];
    char* const ebase = smem + EPI_OFF + wave * (32 * EROW);              // this wave's staging rows
    char* const pbase = smem + EPI_OFF + (wave ^ 4) * (32 * EROW);        // the K-half partner's
    char* const otile = smem + OUT_OFF;

    auto load_x = [&](int t, u32x4_t* xr) {
        const int m = t * BM + spix;
        const uint32_t base = m < a.M ? (uint32_t)((m * P + sslot * 8) * 2) : kOOBf;
        if (DS) {   // block 0 = t2 row, block 1 = block-input row [, block 2 = its lo plane] (64 channels = 128 bytes each)
            xr[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, base, 0, 0);
            xr[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x2, base, 0, 0);
            if (WP3) xr[NX - 1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x2l, base, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) xr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, base, i * 128, 0);
        }
    };
    auto store_x = [&](const u32x4_t* xr, char* buf) {
#pragma unroll
        for (int i = 0; i < NX; ++i) *(u32x4_t*)(buf + i * (BM * 128) + sdst) = xr[i];
    };
    const uint32_t ncol2 = (uint32_t)((n_wave + ecol) * 2);
    auto row_off = [&](int m) { return m < a.M ? (uint32_t)m * (uint32_t)(C4 * 2) + ncol2 : kOOBf; };
    // residual of one 32-pixel strip: TA channel tiles x 2 passes of 16 B per lane
    auto load_res = [&](int t, int j, u32x4_t* r) {
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int pass = 0; pass < 2; ++pass)
                r[i * 2 + pass] = __builtin_amdgcn_raw_buffer_load_b128(
                    rsrc_r, row_off(t * BM + j * 32 + pass * 16 + erow), i * 64, 0);
    };

    u32x4_t xr[NX];
    load_x(tile, xr);
    int cur = 0;
    u32x4_t rres0[TA * 2], rres1[TA * 2];
    if (!DS) load_res(tile, 0, rres0);
    store_x(xr, smem);
    __syncthreads();   // first tile staged, bias tables written
    for (;;) {
        const bool more = tile + per < mt;
        const int next = more ? tile + per : tile;   // last step: a harmless repeat
        load_x(next, xr);
        const int m0 = tile * BM;
        if (!DS) load_res(tile, 1, rres1);
        const char* xb = smem + cur * XBUF;

        // ================= phase A: out = relu(t2 . W3^T + bias3 + res) ================================
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x16_t acc[TA];
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSA; ++ks) {
                const frag_t xf = *(const frag_t*)(xb + (ks >> 2) * (BM * 128) + j * (32 * 128) + lbase +
                                                   (((2 * (ks & 3) + lhi) ^ lswz) << 4));
                // (K block 2 of the paired DS form = the block input's lo plane: the downsample's hi weights once more)
                const int kw = ks < KSW ? ks : ks - 4;
#pragma unroll
                for (int i = 0; i < TA; ++i) {
                    acc[i] = DT::mfma32(w3[i][kw], xf, acc[i]);
                    if (WP3 && ks < KSW) acc[i] = DT::mfma32(w3l[i][ks], xf, acc[i]);
                }
                if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if (!DS) {
#pragma unroll
                for (int q = 0; q < TA * 2; ++q) {   // one wait for the strip's residual, requested long ago
                    if (j == 0) {
                        asm volatile("" : "+v"(rres0[q]));
                    } else {
                        asm volatile("" : "+v"(rres1[q]));
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < TA; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4_t v = {acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
                    *(f32x4_t*)(ebase + lrow * EROW + (8 * g + 4 * lhi) * 4) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const float* const bz = sbias3 + n_wave + i * 32 + ecol;
                const f32x4_t b0 = *(const f32x4_t*)bz, b1 = *(const f32x4_t*)(bz + 4);
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int mrow = pass * 16 + erow;
                    const f32x4_t f0 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4);
                    const f32x4_t f1 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4 + 16);
                    float v[8] = {f0[0] + b0[0], f0[1] + b0[1], f0[2] + b0[2], f0[3] + b0[3],
                                  f1[0] + b1[0], f1[1] + b1[1], f1[2] + b1[2], f1[3] + b1[3]};
                    if (!DS) {
                        const u32x4_t rv = j == 0 ? rres0[i * 2 + pass] : rres1[i * 2 + pass];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float lo, hi;
                            DT::unpack(rv[e], lo, hi);
                            v[2 * e] += lo;
                            v[2 * e + 1] += hi;
                        }
                    }
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    u32x4_t ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = DT::pack(v[2 * e], v[2 * e + 1]);
                    __builtin_amdgcn_raw_buffer_store_b128(ov, rsrc_y, row_off(m0 + j * 32 + mrow), i * 64, 0);
                    ovf.see(ov);
                    // the same 16 bytes into the out-tile (B operand of phase B)
                    const int n = n_wave + i * 32 + ecol, p = j * 32 + mrow;
                    *(u32x4_t*)(otile + (n >> 6) * (BM * 128) + p * 128 + ((((n & 63) >> 3) ^ ((p >> 1) & 7)) << 4)) = ov;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            if (!DS && j == 0) load_res(next, 0, rres0);   // next tile's strip 0, one tile ahead
        }
        __syncthreads();   // (1) the out-tile is complete

        // ================= phase B: t1' = relu(out . W1^T + bias1) ======================================
        {
            constexpr int NJ = WIDE ? 2 : 1;    // strips this wave multiplies
            f32x16_t acc1[NJ];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc1[jj][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSB; ++ks) {
                const int kb = kh * (C4 / 128) + (ks >> 2);
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const int j = WIDE ? jj : jb;
                    const frag_t of = *(const frag_t*)(otile + kb * (BM * 128) + j * (32 * 128) + lbase +
                                                       (((2 * (ks & 3) + lhi) ^ lswz) << 4));
                    acc1[jj] = DT::mfma32(w1[ks], of, acc1[jj]);
                    if (WP1) acc1[jj] = DT::mfma32(w1l[ks], of, acc1[jj]);
                }
                if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            // K halves meet in fp32: P2 = 128: wave (nt, kh) finishes strip kh and hands its partial of the
            // other strip to its partner (w ^ 4); P2 = 64: the kh = 1 wave hands over, kh = 0 finishes.
            const int give = WIDE ? (1 - kh) : 0;
            const bool gives = WIDE || kh == 1;
            const bool finishes = WIDE || kh == 0;
            if (gives) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16_t& s = acc1[WIDE ? give : 0];
                    const f32x4_t v = {s[4 * g + 0], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]};
                    *(f32x4_t*)(ebase + lrow * EROW + (8 * g + 4 * lhi) * 4) = v;
                }
            }
            __syncthreads();   // (2) partials visible
            f32x16_t sum = acc1[WIDE ? kh : 0];
            if (finishes) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4_t v = *(const f32x4_t*)(pbase + lrow * EROW + (8 * g + 4 * lhi) * 4);
                    // fixed order: (K half 0) + (K half 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        sum[4 * g + e] = kh == 0 ? sum[4 * g + e] + v[e] : v[e] + sum[4 * g + e];
                }
            }
            __syncthreads();   // (3) partner done reading before the areas are reused
            if (finishes) {
                const int j = WIDE ? kh : jb;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4_t v = {sum[4 * g + 0], sum[4 * g + 1], sum[4 * g + 2], sum[4 * g + 3]};
                    *(f32x4_t*)(ebase + lrow * EROW + (8 * g + 4 * lhi) * 4) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const float* const bz = sbias1 + nt * 32 + ecol;
                const f32x4_t b0 = *(const f32x4_t*)bz, b1 = *(const f32x4_t*)(bz + 4);
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int mrow = pass * 16 + erow;
                    const f32x4_t f0 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4);
                    const f32x4_t f1 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4 + 16);
                    float v[8] = {f0[0] + b0[0], f0[1] + b0[1], f0[2] + b0[2], f0[3] + b0[3],
                                  f1[0] + b1[0], f1[1] + b1[1], f1[2] + b1[2], f1[3] + b1[3]};
                    if (a.relu2) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    u32x4_t ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = DT::pack(v[2 * e], v[2 * e + 1]);
                    const int m = m0 + j * 32 + mrow;
                    const uint32_t off = m < a.M ? (uint32_t)m * (uint32_t)(P2 * 2) + (uint32_t)((nt * 32 + ecol) * 2) : kOOBf;
                    __builtin_amdgcn_raw_buffer_store_b128(ov, rsrc_y2, off, 0, 0);
                    ovf.see(ov);
                }
            }
        }
        if (!more) break;
        store_x(xr, smem + (cur ^ 1) * XBUF);
        tile = next;
        cur ^= 1;
        __syncthreads();   // (4) next t2 tile staged; out-tile and staging areas free again
    }
    ovf.flush(a.ovf);
}

bool conv_c3c1_admissible(const ConvArgs& a) {
    // a = the conv3 of a bottleneck (1x1 s1) with the following block's conv1 attached; its residual is
    // either a tensor (a.res) or - DS form, planes 64 - the 64-channel block input a.x2, whose
    // downsample weights are concatenated to a.w along K ([Cout][64 + 64]) and biases summed in a.bias
    const bool ds = a.x2 != nullptr;
#ifdef DIR_EXPERIMENTS
    if (a.Cin == 256) return conv_seam3_admissible(a);     // layer3: the streamed-weights form (conv_seam3.hip, experiments builds)
#endif
    return a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.OH && a.W == a.OW &&
           (a.Cin == 64 || a.Cin == 128) && a.Cout == 4 * a.Cin && (ds ? (a.res == nullptr && a.Cin == 64 && a.Cin2 == 64)
                                                                       : a.res != nullptr) &&
           a.w2 != nullptr && a.bias2 != nullptr && a.y2 != nullptr &&
           (a.Cout2 == a.Cin || (!ds && a.Cin == 64 && a.Cout2 == 128)) && (long)a.M * a.Cout < (1L << 30) &&
           // paired weights (DIR_FP16P): layer1's seams; conv3's are pairs whenever anything is, and the block input of the
           // DS form comes with its lo plane exactly then
           ((!a.w_lo && !a.w2_lo && !a.x2_lo) ||
            (a.Cin == 64 && a.w_lo && (a.w2_lo || a.Cout2 == 128) && (ds ? a.x2_lo != nullptr : a.x2_lo == nullptr)));
}

template <class DT, int P, bool DS, int P2 = P, bool WP3 = false, bool WP1 = false>
static hipError_t launch_c3c1(const ConvArgs& a, hipStream_t stream) {
    constexpr int XBUF = 64 * (P + (DS ? 64 : 0) + (DS && WP3 ? 64 : 0)) * 2;
    constexpr int LDS = 2 * XBUF + 8 * 32 * (32 * 4 + 16) + 64 * 4 * P * 2 + (4 * P + P2) * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv_c3c1_kernel<DT, P, DS, P2, WP3, WP1>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.x_bytes = (uint32_t)((size_t)a.M * a.Cin * 2);
    const int mt = (a.M + 63) / 64;
    const int ncu = cu_count();
    hipLaunchKernelGGL(kern, dim3(mt < ncu ? mt : ncu), dim3(512), LDS, stream, b);
    return hipGetLastError();
}

hipError_t conv_c3c1_launch(const ConvArgs& a, int dtype, hipStream_t stream) {
#ifdef DIR_EXPERIMENTS
    if (a.Cin == 256) return conv_seam3_launch(a, dtype, stream);
#endif
    // the DS form with its work split by wave role (conv_c3c1lc.hip, round 6; DIRTORCH_AMD_NO_C3C1LC: the one-role kernel below)
    if (a.x2 && !env().no_c3c1lc && conv_c3c1ds_lc_admissible(a)) return conv_c3c1ds_lc_launch(a, dtype, stream);
    if (a.w_lo) {   // DIR_FP16P: conv3's (+ downsample's) weights are pairs; conv1's are when they belong to layer1 too
        if (dtype != DIR_FP16) return hipErrorInvalidValue;
        if (a.x2) return a.w2_lo ? launch_c3c1<FP16, 64, true, 64, true, true>(a, stream) : hipErrorInvalidValue;
        if (a.Cout2 == 128)
            return a.w2_lo ? launch_c3c1<FP16, 64, false, 128, true, true>(a, stream)
                           : launch_c3c1<FP16, 64, false, 128, true, false>(a, stream);
        return a.w2_lo ? launch_c3c1<FP16, 64, false, 64, true, true>(a, stream) : hipErrorInvalidValue;
    }
    if (a.x2)
        return dtype == DIR_BF16 ? launch_c3c1<BF16, 64, true>(a, stream) : launch_c3c1<FP16, 64, true>(a, stream);
    if (a.Cin == 128)
        return dtype == DIR_BF16 ? launch_c3c1<BF16, 128, false>(a, stream) : launch_c3c1<FP16, 128, false>(a, stream);
    if (a.Cout2 == 128)   // layer1 -> layer2
        return dtype == DIR_BF16 ? launch_c3c1<BF16, 64, false, 128>(a, stream) : launch_c3c1<FP16, 64, false, 128>(a, stream);
    return dtype == DIR_BF16 ? launch_c3c1<BF16, 64, false>(a, stream) : launch_c3c1<FP16, 64, false>(a, stream);
}

}  // namespace dir
