// conv_igemm.hip — implicit-GEMM convolution for gfx950 (MFMA 32x32x16, 16-bit in / fp32 acc).
//
// Replaces every Conv2d + BatchNorm2d(eval) (+ residual add) (+ ReLU) of the reference trunk
// (dirtorch/nets/backbones/resnet.py:56-63,67-87,115-118,136-141) with one fused kernel family.
//
// GEMM view (per launch):   Y[m][n] = act( sum_k X[m][k] * Wt[n][k] + bias[n] (+ res[m][n]) )
//   m = (b, oh, ow) output pixel, n = output channel, k = (r, s, c) filter tap x input channel.
//   X is never materialised: for K-step (r, s, c0..c0+63) the 64-channel slice of input pixel
//   (oh*stride + r - pad, ow*stride + s - pad) is one contiguous 128-byte run of the NHWC tensor.
//   The stem (7x7 s2, Cin = 3) arrives as a 4x4 s1 convolution over the 2x2 space-to-depth image
//   (Cin = 16): one K-step = one filter row = four neighbouring pixels = the same 128-byte run.
//
// Data movement: HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) through two buffer
//   descriptors (activations, weights).  Per lane the 32-bit byte offset is computed ONCE; the
//   K-step advances through the instruction's scalar offset, so a 1x1 convolution spends no
//   VALU work per load.  Zero padding and the ragged last pixel tile use the descriptor's
//   bounds check: an out-of-range offset makes the hardware write zeros.
// LDS: an NST-slot ring of stages; a stage = X-tile [BM][64] + W-tile [BN][64], rows of 128 B whose
//   16-byte chunks are XOR-swizzled with ((row >> 1) & 7) so the MFMA fragment reads (ds_read_b128:
//   row = lane & 31, chunk = 2*ks + lane/32) are bank-conflict free.  The DMA destination is
//   lane-linear, so the swizzle is applied to the per-lane SOURCE chunk.  Waits are counted
//   (vmcnt = loads per stage x stages still in flight) and the barrier is the raw s_barrier, so
//   later stages stay in flight across it.
// MFMA operand roles are swapped (A = weights, B = pixels) so each lane ends up holding 4
//   consecutive output channels of one pixel.  Accumulators start at the folded-BN bias; the
//   epilogue stages them through LDS (fp32) and emits whole 16-byte runs (8 channels) with
//   residual add / ReLU / v_cvt_pk conversion fused.  Residual tiles are fetched before the K loop.
#include "dir_common.h"
#include "conv_igemm.h"

#include <stdlib.h>
#include <string.h>
#include <algorithm>

namespace dir {

// voffset beyond any descriptor (tensors are < 2^31 bytes): the DMA writes zeros.  2^31 cannot wrap
// in 32 bits when the scalar K offset is added, whichever way the bounds check treats soffset.
static constexpr uint32_t kOOB = 0x80000000u;

// 16 bytes per lane, global/L2 -> LDS at (wave-uniform `lds`) + lane * 16; `soff` rides in an SGPR.
// (Kept in a __device__ function: used directly inside the kernel template's lambda, hipcc 7.2
// silently drops the kernel's host stub.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)lds, 16, voff, soff, 0, 0);
}

__device__ inline uint32_t fast_div(uint32_t n, uint32_t mul, uint32_t shr) {
    return mul ? (__umulhi(n, mul) >> shr) : n;  // mul == 0 encodes division by 1
}

// SPLITK instantiations (a few small-tile variants) carry the split-K bookkeeping; the others compile
// exactly as if it did not exist - its extra scalar state costs 8-70 VGPRs in the big tiles.
// (The two-source form of rounds 2-3 - conv3 + the block's 1x1 downsample as one GEMM over [t2 ; x_s], one tile per
// workgroup - lives on the persistent ring since round 4: conv_persist.hip DUAL, bit-identical, 5-13 % faster.)
template <class DT, int BM, int BN, int WGM, int WGN, int NST, int BK, bool CIN16, bool SPLITK>
__global__ void __launch_bounds__(64 * WGM * WGN) conv_igemm_kernel(const ConvArgs a) {
    static_assert(NST >= 2 && NST <= 8, "ring depth");
    static_assert((BK == 64 || BK == 32) && (!CIN16 || BK == 64), "K-step");
    constexpr int RB = BK * 2;    // bytes per LDS row (one pixel / one output channel, BK channels)
    constexpr int CPR = BK / 8;   // 16-byte chunks per row
    constexpr int KS = BK / 16;   // MFMA k-substeps per stage
    constexpr int NT = 64 * WGM * WGN;
    constexpr int TM = BM / WGM / 32;  // pixel (B-operand) tiles per wave
    constexpr int TN = BN / WGN / 32;  // channel (A-operand) tiles per wave
    constexpr int NA = BM * CPR / NT;  // 16-byte X chunks per lane per stage
    constexpr int NB = BN * CPR / NT;  // 16-byte W chunks per lane per stage
    constexpr int LPS = NA + NB;       // DMA instructions per lane per stage
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    static_assert(NA >= 1 && NB >= 1 && (BM * CPR) % NT == 0 && (BN * CPR) % NT == 0, "chunk split");
    static_assert((NT / CPR) % 16 == 0, "swizzle term must be constant per lane");
    constexpr int XS = BM * RB;  // bytes of the X tile of one stage
    constexpr int STAGE_BYTES = (BM + BN) * RB;
    constexpr int EROW = TN * 128 + 16;  // epilogue: one pixel row of TN*32 fp32 + pad
    typedef typename DT::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WGN;
    const int wm = wave / WGN;
    const int lrow = lane & 31;
    const int lhi = lane >> 5;

    const int ntiles = a.tiles_m * a.tiles_n;
    const int wg = SPLITK ? xcd_remap(blockIdx.x % ntiles, ntiles) : xcd_remap(blockIdx.x, gridDim.x);
    const int kz = SPLITK ? blockIdx.x / ntiles : 0;  // split-K slice of this workgroup
    const int tile_n = wg % a.tiles_n;   // n fastest: blocks sharing an X tile run on one XCD
    const int tile_m = wg / a.tiles_n;

    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

    // ---- per-lane source offsets (constant over the K loop) --------------------------------------
    // chunk slot `tid % CPR` of LDS row `tid / CPR (+ i*NT/CPR)` holds source chunk slot ^ swz(row):
    // swz = (row >> 1) & 7 for 128-byte rows, (row >> 2) & 3 for 64-byte rows (one 256-byte bank row
    // holds 2 resp. 4 LDS rows) - both reduce to bits of tid because NT/CPR is a multiple of 16.
    const int srcchunk = (tid & (CPR - 1)) ^ ((tid >> 4) & (CPR - 1));
    const bool one_tap = (a.R * a.S == 1);  // 1x1: no padding, K advances through the scalar offset
    int xbase[NA];       // byte offset of tap (0,0), channel chunk `srcchunk` (may be negative)
    uint32_t xmask[NA];  // bit per tap (stem: per filter row): tap inside the image and row valid
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = i * (NT / CPR) + tid / CPR;
        const int m = tile_m * BM + row;
        const bool mvalid = m < a.M;
        if (a.flat) {  // 1x1, stride 1: output pixel m reads input pixel m
            xbase[i] = (m * a.Cin + srcchunk * 8) * 2;
            xmask[i] = mvalid ? 1u : 0u;
        } else {
            const uint32_t mm = mvalid ? (uint32_t)m : 0u;
            const uint32_t b = fast_div(mm, a.div_ohw_mul, a.div_ohw_shr);
            const uint32_t rem = mm - b * (uint32_t)(a.OH * a.OW);
            const uint32_t oh = fast_div(rem, a.div_ow_mul, a.div_ow_shr);
            const uint32_t ow = rem - oh * (uint32_t)a.OW;
            const int ih0 = (int)oh * a.stride - a.pad;
            const int iw0 = (int)ow * a.stride - a.pad;
            xbase[i] = (((int)b * a.H + ih0) * a.W + iw0) * a.Cin * 2 + srcchunk * 16;
            // validity bits in closed form (R, S <= 4): rows r in [max(0,-ih0), min(R, H-ih0)),
            // columns s in [max(0,-iw0), min(S, W-iw0))
            auto range_bits = [](int lo, int hi) -> uint32_t {
                return hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
            };
            const int R_ = CIN16 ? 4 : a.R;
            const uint32_t rbits = range_bits(max(0, -ih0), min(R_, a.H - ih0));
            if (CIN16) {
                // one K-step per filter row; this lane's chunk is pixel (srcchunk >> 1) of the row
                const bool wok = (unsigned)(iw0 + (srcchunk >> 1)) < (unsigned)a.W;
                xmask[i] = (mvalid && wok) ? rbits : 0u;
            } else {
                const uint32_t cbits = range_bits(max(0, -iw0), min(a.S, a.W - iw0));
                uint32_t mask = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) mask |= ((rbits >> r) & 1u) ? (cbits << (r * a.S)) : 0u;
                xmask[i] = mvalid ? mask : 0u;
            }
        }
    }
    uint32_t xvoff[NA];  // 1x1 path: final voffset with the row mask folded in
#pragma unroll
    for (int i = 0; i < NA; ++i) xvoff[i] = (xmask[i] & 1u) ? (uint32_t)xbase[i] : kOOB;
    uint32_t wvoff[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = i * (NT / CPR) + tid / CPR;
        wvoff[i] = (uint32_t)(((tile_n * BN + row) * a.Ktot + srcchunk * 8) * 2);
    }
    // K-step t: filter tap `tap` (stem: filter row), byte offset `koff` of that tap/channel slice
    auto issue = [&](int t, int tap, int koff, char* stage) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            char* dst = stage + (i * NT + wave * 64) * 16;
            if (one_tap) {
                dma16(rsrc_x, dst, xvoff[i], koff);
            } else {
                const uint32_t v = ((xmask[i] >> tap) & 1u) ? (uint32_t)(xbase[i] + koff) : kOOB;
                dma16(rsrc_x, dst, v, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            dma16(rsrc_w, stage + XS + (i * NT + wave * 64) * 16, wvoff[i], t * RB);
    };

    // ---- fragment read offsets -----------------------------------------------------------------
    const int lswz = BK == 64 ? ((lane >> 1) & 7) : ((lane >> 2) & 3);
    int loff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) loff[ks] = lrow * RB + (((2 * ks + lhi) ^ lswz) << 4);
    const int xfrag = (wm * TM * 32) * RB;
    const int wfrag = XS + (wn * TN * 32) * RB;

    // ---- accumulators start at the bias of their 4 consecutive channels -----------------------
    f32x16_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4_t b4 = {0.f, 0.f, 0.f, 0.f};   // split-K partial sums carry no bias
            if (!SPLITK)
                b4 = *(const DIR_GLOBAL f32x4_t*)(a.bias + tile_n * BN + wn * TN * 32 + i * 32 + 8 * g +
                                                  4 * lhi);
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = b4[e];
        }

    auto compute = [&](const char* stage) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            frag_t wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i)
                wf[i] = *(const frag_t*)(stage + wfrag + i * 32 * RB + loff[ks]);
#pragma unroll
            for (int j = 0; j < TM; ++j)
                xf[j] = *(const frag_t*)(stage + xfrag + j * 32 * RB + loff[ks]);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = DT::mfma32(wf[i], xf[j], acc[i][j]);
        }
    };

    // ---- residual tile fetched up front: its HBM latency hides under the whole K loop ----------
    constexpr int LPR = TN * 4;                  // lanes covering one pixel row (8 channels each)
    constexpr int RPP = 64 / LPR;                // pixel rows per pass
    constexpr int NPASS = 32 / RPP;              // passes per 32-pixel strip
    constexpr bool PRE_RES = (TM * NPASS <= 16); // 16 B per lane each: at most 64 VGPRs
    const int ecol = (lane % LPR) * 8;
    const int erow = lane / LPR;
    const int n_glob = tile_n * BN + wn * TN * 32 + ecol;
    const int m_epi = tile_m * BM + wm * TM * 32;
    u32x4_t rres[PRE_RES ? TM : 1][PRE_RES ? NPASS : 1];
    if (PRE_RES && !SPLITK && a.res) {
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int m = m_epi + j * 32 + pass * RPP + erow;
                const int mc = m < a.M ? m : 0;  // clamped rows are never stored
                rres[PRE_RES ? j : 0][PRE_RES ? pass : 0] =
                    gload16(a.res + ((size_t)mc * a.Cout + n_glob));
            }
    }

    // ---- K loop: NST-slot ring fed by LDS-DMA ----------------------------------------------------
    // Stages t+1 .. t+NST-1 stay in flight while stage t is consumed.  One barrier per K-step:
    // passing it means (a) stage t has landed for every wave (each waited on its own counted vmcnt
    // first), (b) every wave has finished reading slot (t-1) % NST, the slot refilled right after.
    const int cpb = CIN16 ? 1 : (a.Cin / BK);  // K-steps per filter tap
    const int t_begin = SPLITK ? (int)((long)a.T * kz / a.ksplit) : 0;
    const int T = SPLITK ? (int)((long)a.T * (kz + 1) / a.ksplit) - t_begin : a.T;  // this workgroup's K-steps
    // K order: channel slice outermost, filter taps innermost.  Consecutive K-steps of a 3x3 conv
    // then read the SAME 64-channel slice of the input at pixel-shifted positions, so eight of the
    // nine tap reads hit the XCD's L2 (a tap-major order spaced those re-reads Cin/64 steps apart,
    // 32 CUs x that footprint overflowed the 4 MB L2 and the 3x3 layers fetched their input ~3x
    // from the fabric - profiles/r01_bench_b32_hbm_pmc.json history).  The stem (CIN16) has one
    // K-step per filter row and keeps row order.
    int tap = 0, cc = 0, r = 0, s = 0;         // state of the step being ISSUED
    if (SPLITK && t_begin > 0) {               // split-K: start in the middle of the K sequence
        if (CIN16) {
            r = tap = t_begin;
        } else {
            const int taps = a.R * a.S;
            cc = t_begin / taps;
            tap = t_begin - cc * taps;
            r = tap / a.S;
            s = tap - r * a.S;
        }
    }
    auto koff_now = [&]() {                    // bytes into the input, relative to tap (0,0)
        return CIN16 ? (r * a.W * 32) : (((r * a.W + s) * a.Cin + cc * BK) * 2);
    };
    auto wstep_now = [&]() {                   // index of this K-step's slice in a weight row
        return CIN16 ? r : (tap * cpb + cc);
    };
    auto advance = [&]() {
        if (CIN16) {
            ++r;
            ++tap;
        } else {
            ++tap;
            if (++s == a.S) {
                s = 0;
                if (++r == a.R) {
                    r = 0;
                    tap = 0;
                    ++cc;
                }
            }
        }
    };
    int issued = 0;
#pragma unroll
    for (int p = 0; p < NST - 1; ++p) {
        if (p < T) {
            issue(wstep_now(), tap, koff_now(), smem + p * STAGE_BYTES);
            advance();
            ++issued;
        }
    }
    int slot_c = 0;        // slot holding stage t
    int slot_i = issued;   // slot the next issue goes to
    if (slot_i == NST) slot_i = 0;
    for (int t = 0; t < T; ++t) {
        const int ahead = issued - 1 - t;  // stages issued beyond t: 0 .. NST-2
        // (deep rings, round 6: the small-tile variants of the small-map regime keep up to six stages in flight)
        static_assert((NST - 2) * LPS <= 63, "vmcnt immediate");
        if (NST >= 8 && ahead >= 6) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NST >= 8 ? 6 * LPS : 0) : "memory");
        } else if (NST >= 7 && ahead >= 5) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NST >= 7 ? 5 * LPS : 0) : "memory");
        } else if (NST >= 6 && ahead >= 4) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NST >= 6 ? 4 * LPS : 0) : "memory");
        } else if (NST >= 5 && ahead >= 3) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NST >= 5 ? 3 * LPS : 0) : "memory");
        } else if (NST >= 4 && ahead >= 2) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * LPS) : "memory");
        } else if (NST >= 3 && ahead == 1) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ring_barrier();
        if (issued < T) {
#ifndef DIR_EXP_NO_FILL     // experiment builds (scripts/exp_abl.sh conv_igemm DIR_EXP_NO_FILL 1): MFMA + fragment reads alone
            issue(wstep_now(), tap, koff_now(), smem + slot_i * STAGE_BYTES);
#endif
            advance();
            ++issued;
            if (++slot_i == NST) slot_i = 0;
        }
#ifndef DIR_EXP_FILL_ONLY   // experiment builds: the LDS-DMA ring alone
        compute(smem + slot_c * STAGE_BYTES);
#endif
        if (++slot_c == NST) slot_c = 0;
    }
    __syncthreads();  // all fragment reads done before the epilogue reuses the ring
    Ovf<DT> ovf;

    // ---- epilogue: acc -> LDS (fp32, pixel-major) -> residual / ReLU / convert -> 16-byte stores --
    char* ebase = smem + wave * (32 * EROW);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4_t v = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2],
                             acc[i][j][4 * g + 3]};
                const int n_local = i * 32 + 8 * g + 4 * lhi;
                *(f32x4_t*)(ebase + lrow * EROW + n_local * 4) = v;
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
            if (SPLITK && m < a.M) {   // raw fp32 partial sums; conv_splitk_finalize does the rest
                float* po = a.partial + ((size_t)kz * a.M + m) * a.Cout + n_glob;
                *(DIR_GLOBAL f32x4_t*)po = f0;
                *(DIR_GLOBAL f32x4_t*)(po + 4) = f1;
            } else if (!SPLITK && m < a.M) {
                float v[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
                const size_t o = (size_t)m * a.Cout + n_glob;
                if (a.res) {
                    const u32x4_t rv = PRE_RES ? rres[PRE_RES ? j : 0][PRE_RES ? pass : 0]
                                               : gload16(a.res + o);
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
                gstore16(a.y + o, ov);
                ovf.see(ov);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    ovf.flush(a.ovf);
}

// ---- variant table ----------------------------------------------------------------------------
static void fastdiv_init(uint32_t d, uint32_t& mul, uint32_t& shr) {
    // q = umulhi(n, mul) >> shr is exact for 0 <= n < 2^31 (mul = ceil(2^(31+l) / d), l = ceil(log2 d))
    if (d <= 1) {
        mul = 0;
        shr = 0;
        return;
    }
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    const uint32_t p = 31 + l;
    mul = (uint32_t)(((1ull << p) + d - 1) / d);
    shr = p - 32;
}

template <class DT, int BM, int BN, int WGM, int WGN, int NST, int BK, bool CIN16, bool SPLITK = false>
static hipError_t launch_variant(const ConvArgs& a, hipStream_t stream) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int TN = BN / WGN / 32;
    constexpr int EROW = TN * 128 + 16;
    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    constexpr int EPI_BYTES = (NT / 64) * 32 * EROW;
    constexpr int LDS = (NST * STAGE_BYTES > EPI_BYTES) ? NST * STAGE_BYTES : EPI_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv_igemm_kernel<DT, BM, BN, WGM, WGN, NST, BK, CIN16, SPLITK>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.T = a.Ktot / BK;
    b.tiles_m = ceil_div(a.M, BM);
    b.tiles_n = a.Cout / BN;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    b.flat = (a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.OH && a.W == a.OW);
    fastdiv_init((uint32_t)(a.OH * a.OW), b.div_ohw_mul, b.div_ohw_shr);
    fastdiv_init((uint32_t)a.OW, b.div_ow_mul, b.div_ow_shr);
    // a short K loop never touches the far slots of the ring: ask for less LDS, more residency
    const int used = (b.T < NST ? b.T : NST) * STAGE_BYTES;
    const int lds = used > EPI_BYTES ? used : EPI_BYTES;
    const int nz = SPLITK ? a.ksplit : 1;
    hipLaunchKernelGGL(kern, dim3(b.tiles_m * b.tiles_n * nz), dim3(NT), lds, stream, b);
    return hipGetLastError();
}

#define DIR_VARIANT(BM, BN, WGM, WGN, NST, BK, NAME)                                         \
    {NAME, BM, BN, 64 * WGM * WGN, NST, BK,                                                  \
     {launch_variant<BF16, BM, BN, WGM, WGN, NST, BK, false>,                                \
      launch_variant<FP16, BM, BN, WGM, WGN, NST, BK, false>},                               \
     {nullptr, nullptr}, 0, {nullptr, nullptr}, {nullptr, nullptr}}
// ... plus the split-K instantiation (small-M layers)
#define DIR_VARIANT_SK(BM, BN, WGM, WGN, NST, BK, NAME)                                      \
    {NAME, BM, BN, 64 * WGM * WGN, NST, BK,                                                  \
     {launch_variant<BF16, BM, BN, WGM, WGN, NST, BK, false>,                                \
      launch_variant<FP16, BM, BN, WGM, WGN, NST, BK, false>},                               \
     {nullptr, nullptr}, 0,                                                                  \
     {launch_variant<BF16, BM, BN, WGM, WGN, NST, BK, false, true>,                          \
      launch_variant<FP16, BM, BN, WGM, WGN, NST, BK, false, true>}, {nullptr, nullptr}}
// BN == 64 variants also carry the Cin == 16 (space-to-depth stem) instantiation.
#define DIR_VARIANT16(BM, BN, WGM, WGN, NST, NAME)                                           \
    {NAME, BM, BN, 64 * WGM * WGN, NST, 64,                                                  \
     {launch_variant<BF16, BM, BN, WGM, WGN, NST, 64, false>,                                \
      launch_variant<FP16, BM, BN, WGM, WGN, NST, 64, false>},                               \
     {launch_variant<BF16, BM, BN, WGM, WGN, NST, 64, true>,                                 \
      launch_variant<FP16, BM, BN, WGM, WGN, NST, 64, true>}, 0, {nullptr, nullptr}, {nullptr, nullptr}}

// name = <pixels>x<channels>_w<waves m>x<waves n>[_s<ring depth>][_k<K-step>]
static const ConvVariant kVariants[] = {
    DIR_VARIANT_SK(128, 128, 2, 2, 2, 64, "128x128_w2x2"),
    DIR_VARIANT16(128, 64, 2, 2, 2, "128x64_w2x2"),
    DIR_VARIANT16(256, 64, 4, 1, 2, "256x64_w4x1"),
    DIR_VARIANT(256, 128, 4, 2, 2, 64, "256x128_w4x2"),
    DIR_VARIANT(128, 256, 2, 4, 2, 64, "128x256_w2x4"),
    DIR_VARIANT(256, 256, 4, 2, 2, 64, "256x256_w4x2"),
    DIR_VARIANT_SK(64, 128, 2, 2, 2, 64, "64x128_w2x2"),
    DIR_VARIANT16(64, 64, 2, 1, 2, "64x64_w2x1"),
    DIR_VARIANT(64, 128, 2, 2, 4, 64, "64x128_w2x2_s4"),
    DIR_VARIANT(128, 128, 2, 2, 3, 64, "128x128_w2x2_s3"),
    DIR_VARIANT16(128, 64, 2, 2, 4, "128x64_w2x2_s4"),
    DIR_VARIANT(256, 128, 4, 2, 3, 64, "256x128_w4x2_s3"),
    DIR_VARIANT(128, 256, 2, 4, 3, 64, "128x256_w2x4_s3"),
    DIR_VARIANT16(256, 64, 4, 1, 3, "256x64_w4x1_s3"),
    DIR_VARIANT(256, 256, 4, 2, 4, 32, "256x256_w4x2_s4_k32"),
    DIR_VARIANT(256, 256, 4, 2, 3, 32, "256x256_w4x2_s3_k32"),
    DIR_VARIANT(256, 128, 4, 2, 4, 32, "256x128_w4x2_s4_k32"),
    DIR_VARIANT(128, 128, 2, 2, 4, 32, "128x128_w2x2_s4_k32"),
    // 72 KB of LDS and <= 128 VGPRs: two 512-thread workgroups share a CU, so one's epilogue
    // overlaps the other's K loop (the memory-bound 1x1 convs of layer3/4)
    DIR_VARIANT(256, 128, 4, 2, 3, 32, "256x128_w4x2_s3_k32"),
    DIR_VARIANT(128, 256, 2, 4, 3, 32, "128x256_w2x4_s3_k32"),
    DIR_VARIANT(128, 256, 2, 2, 3, 32, "128x256_w2x2_s3_k32"),   // two 256-thread workgroups per CU
    DIR_VARIANT(128, 128, 2, 2, 3, 32, "128x128_w2x2_s3_k32"),   // 48 KB: three workgroups per CU
    // 16 waves of 64x64 on one CU (128 VGPRs): more fragment reads in flight under the matrix
    // pipe - the 3x3 convs of layer3/4
    DIR_VARIANT(256, 256, 4, 4, 2, 64, "256x256_w4x4"),
    // the small-map regime (batch 1 / 224^2 buckets: 1 000 - 13 000 pixels per layer): 64x64 tiles give every CU one even at
    // 4 096 pixels x 256 channels.  _s4 (64 KB: two workgroups per CU - what forwards overlapping on several streams need) is the
    // picker's choice: batch 1 at 1024^2 768 -> 799 img/s on one stream, 1 320 -> 1 366 on four; _s8 (six 16 KB stages in flight,
    // 128 KB) is 1 % better on one stream and 20 % worse on four (806 / 1 075, gpurun_out/r6smallab2) - a tuner candidate
    DIR_VARIANT_SK(64, 64, 2, 2, 8, 64, "64x64_w2x2_s8"),
    DIR_VARIANT_SK(64, 64, 2, 2, 4, 64, "64x64_w2x2_s4"),
    // 3x3 stride-1 from an LDS-resident input patch (conv_patch.hip): 8x32 pixels x all channels
    {"256x64_patch3x3", 256, 64, 256, 3, 64, {nullptr, nullptr}, {nullptr, nullptr}, 1, {nullptr, nullptr}, {nullptr, nullptr}},
    {"256x128_patch3x3", 256, 128, 512, 3, 64, {nullptr, nullptr}, {nullptr, nullptr}, 1, {nullptr, nullptr}, {nullptr, nullptr}},
    // ... 64 -> 64 channels without a residual: the whole filter resident in LDS, double-buffered patches, loader waves
    // fetch while consumer waves multiply (conv_patchlc.hip)
    {"256x64_patchlc3x3", 256, 64, 512, 2, 64, {nullptr, nullptr}, {nullptr, nullptr}, 8, {nullptr, nullptr}, {nullptr, nullptr}},
    // ... for the wide 3x3 layers (256 / 512 channels): the patch one 64-channel plane at a time, Cout tiled by 256
    {"256x256_patch3x3s", 256, 256, 512, 2, 64, {nullptr, nullptr}, {nullptr, nullptr}, 5, {nullptr, nullptr}, {nullptr, nullptr}},
    // ... 512 pixels x 128 channels per workgroup, 32-channel planes double-buffered, one filter row per weight stage
    {"512x128_patch3x3w", 512, 128, 512, 3, 32, {nullptr, nullptr}, {nullptr, nullptr}, 6, {nullptr, nullptr}, {nullptr, nullptr}},
    // 3x3 STRIDE 2 (conv2 of the first block of layers 2-4): persistent, 8 x 32 output pixels x 128 channels from a 17 x 65 patch,
    // 32-channel planes double-buffered with even / odd input columns apart, weights straight into registers (conv_patchs2.hip)
    {"256x128_patchs2", 256, 128, 768, 2, 32, {nullptr, nullptr}, {nullptr, nullptr}, 12, {nullptr, nullptr}, {nullptr, nullptr}},
    // persistent workgroups, next tile's first K-stage issued before the epilogue (conv_persist.hip)
    {"256x256_persist1x1", 256, 256, 512, 2, 64, {nullptr, nullptr}, {nullptr, nullptr}, 2, {nullptr, nullptr}, {nullptr, nullptr}},
    // the same with three K-steps of the pixel operand in the ring (HBM requests in flight: 32 -> 64+ KB per CU)
    // ... and its two-source form (conv3 + downsample of the first block of layers 2-4)
    {"256x256_persist1x1_x3", 256, 256, 512, 3, 64, {nullptr, nullptr}, {nullptr, nullptr}, 4, {nullptr, nullptr},
     {conv1x1_persist_dual_bf16, conv1x1_persist_dual_fp16}},
    // persistent 128x256 tile, loader waves feed ONE three-slot K ring over all the tiles of a workgroup, consumer waves
    // multiply; 1x1 convs without a residual (conv_ring.hip)
#ifdef DIR_EXPERIMENTS   // (csrc/build.sh with DIR_EXPERIMENTS=1: ties conv_persist.hip inside the network, not a default build's kernel)
    {"128x256_ring1x1", 128, 256, 512, 3, 64, {nullptr, nullptr}, {nullptr, nullptr}, 7, {nullptr, nullptr}, {nullptr, nullptr}},
#endif
    // persistent, 64 output channels x K <= 256 per wave held in VGPRs, only pixels stream (conv_wreg.hip)
    {"64x512_wreg1x1", 64, 512, 512, 2, 64, {nullptr, nullptr}, {nullptr, nullptr}, 3, {nullptr, nullptr}, {nullptr, nullptr}},
    // small maps (batch 1 at native size): 64 x 64 tiles, four consumer + four loader waves that meet on LDS counters, no workgroup
    // barrier in the loop (conv_small.hip); _s4 = 64 KB (two workgroups per CU), _s8 = 128 KB
    {"64x64_small_s8", 64, 64, 512, 8, 64, {nullptr, nullptr}, {nullptr, nullptr}, 11, {nullptr, nullptr}, {nullptr, nullptr}},
    {"64x64_small_s4", 64, 64, 512, 4, 64, {nullptr, nullptr}, {nullptr, nullptr}, 11, {nullptr, nullptr}, {nullptr, nullptr}},
    {"64x64_small_s4k2", 64, 64, 512, 5, 64, {nullptr, nullptr}, {nullptr, nullptr}, 11, {nullptr, nullptr}, {nullptr, nullptr}},   // 4 slots x 2 K-steps per stage
    // the deep-X ring with loader / consumer wave roles (eight consumers, four loaders): 1x1 without a residual, and its two-source form
    {"256x256_lc1x1", 256, 256, 768, 3, 64, {nullptr, nullptr}, {nullptr, nullptr}, 10, {nullptr, nullptr},
     {conv1x1_lc_dual_bf16, conv1x1_lc_dual_fp16}},
    // the two-source GEMM of layer2's first block (K = 128 + 256) with 32 output channels x 384 inputs per wave held in VGPRs,
    // 256 channels per workgroup (conv_wregd.hip); a launch_dual-only entry
    {"64x256_wregd1x1", 64, 256, 512, 2, 64, {nullptr, nullptr}, {nullptr, nullptr}, 9, {nullptr, nullptr},
     {conv1x1_wregd_bf16, conv1x1_wregd_fp16}},
};
static constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

int conv_variant_count() { return kNumVariants; }
const ConvVariant& conv_variant(int i) { return kVariants[i]; }

bool conv_variant_admissible(int v, const ConvArgs& a) {
    if (v < 0 || v >= kNumVariants) return false;
    const ConvVariant& cv = kVariants[v];
    if (cv.kind == 1) return a.Cout == cv.BN && conv_patch3x3_admissible(a);
    if (cv.kind == 5) return conv_patch3x3s_admissible(a);
    if (cv.kind == 6) return conv_patch3x3w_admissible(a);
    if (cv.kind == 2) return conv1x1_persist_admissible(a);
    if (cv.kind == 4) return conv1x1_persist_admissible(a) && a.res == nullptr;   // the deep-X form has no residual path
    if (cv.kind == 3) return conv1x1_wreg_admissible(a);
#ifdef DIR_EXPERIMENTS
    if (cv.kind == 7) return conv1x1_ring_admissible(a);
#endif
    if (cv.kind == 8) return conv_patch64_lc_admissible(a);
    if (cv.kind == 12) return conv_patch3x3s2_admissible(a);
    if (cv.kind == 11) return conv_small_admissible(a) && (cv.stages != 5 || (a.Ktot / 64) % 2 == 0);
    if (cv.kind == 10) return a.x2 == nullptr && conv1x1_lc_admissible(a);
    if (cv.kind == 9) return conv1x1_wregd_admissible(a);   // (two-source shapes only: never true for a plain conv)
    if (a.Cout % cv.BN != 0) return false;
    if (a.Cin == 16 && cv.launch16[0] == nullptr) return false;
    return true;
}

int conv_splitk_factor(int v, const ConvArgs& a);

static int find_variant(const char* name) {
    for (int v = 0; v < kNumVariants; ++v)
        if (strcmp(kVariants[v].name, name) == 0) return v;
    return -1;
}

// Heuristic used when a shape has not been autotuned (variable-size images, the reference's real
// workload).  Distilled from the autotuner's choices on ResNet-101 at 1024^2 (profiles/
// r01_tuned_variants_b32_1024.txt):
//   * narrow 3x3 layers (Cin 64) take the LDS-patch kernel, K = 64 layers the long 256x64 tile;
//   * wide outputs with a real K loop (layer3/4 conv1 + conv2) a 256x256 tile: the persistent
//     kernel for 1x1, the 16-wave form for 3x3;
//   * wide outputs with a short K loop (the 256 -> 1024 conv3 of layer3, residual + store bound)
//     and everything with 128/512 outputs (layer2) the 72 KB / 128-VGPR tiles that fit two
//     workgroups per CU, so that one's epilogue overlaps the other's K loop;
// each falling back to smaller pixel tiles until about three quarters of the CU slots get a tile.
int conv_pick_variant(const ConvArgs& a) {
    {
        // layer1's 64 -> 64 3x3: filter resident in LDS, loader / consumer waves (DIRTORCH_AMD_NO_PATCHLC: A/B and bisecting)
        const bool no_lc = env().no_patchlc;
        const int v = find_variant("256x64_patchlc3x3");
        if (!no_lc && v >= 0 && conv_variant_admissible(v, a) &&
            (long)a.B * ((a.OH + 7) / 8) * ((a.OW + 31) / 32) >= 192)
            return v;
    }
    {
        // 3x3 stride 2 over 128 channels (layer2.0's conv2, bound by its full-resolution input): the patch kernel with even / odd
        // column runs - 0.222 -> 0.208 ms standalone at batch 32 (gpurun_out/r6s2b).  Not for the 256 / 512-channel ones of
        // layer3.0 / 4.0: those are matrix-bound, and one plane in flight per CU costs them 15-25 us against the 16-wave tile (0.174
        // vs 0.149, 0.152 vs 0.135 ms) - the variant stays a tuner candidate there.  DIRTORCH_AMD_NO_PATCHS2: generic tiles again.
        const int v = find_variant("256x128_patchs2");
        if (!env().no_patchs2 && v >= 0 && a.Cin <= 128 && conv_variant_admissible(v, a) &&
            (long)a.B * ((a.OH + 7) / 8) * ((a.OW + 31) / 32) * (a.Cout / 128) >= 192)
            return v;
    }
    for (int v = 0; v < kNumVariants; ++v)
        if (kVariants[v].kind == 1 && a.Cin == 64 && conv_variant_admissible(v, a)) return v;
    // the residual 1x1 convs with K <= 256 (layer2/3 conv3): weights stationary in registers, as long
    // as every persistent workgroup gets at least ~4 pixel tiles to amortise loading them
    {
        const bool no_wreg = env().no_wreg;   // A/B and bisecting
        const int v = find_variant("64x512_wreg1x1");
        if (!no_wreg && v >= 0 && conv_variant_admissible(v, a) &&
            (long)ceil_div(a.M, 64) * (a.Cout / 512) >= 1024)
            return v;
    }
    const int T = a.Ktot / 64;
    struct Cand { const char* name; int wg_per_cu; };
    Cand c[12];
    int n = 0;
    {
        // 3x3 stride 1 over >= 128 channels (conv2 of layer2 / 3 / 4): 512 pixels x 128 channels per workgroup with
        // double-buffered 32-channel planes (conv_patchw.hip) - A/B at batch 32 (gpurun_out/pw): layer2 175 -> 153 us,
        // layer3 133 -> 124 us, layer4 121 -> 113 us.  Falls through (like every candidate) when it is not admissible
        // or leaves CUs without a tile.
        const bool no_pw = env().no_patchw;   // A/B and bisecting
        if (!no_pw && a.R * a.S > 1 && a.Cin >= 128) c[n++] = {"512x128_patch3x3w", 1};
    }
    if (T <= 1 || a.Cout % 128 != 0) {
        c[n++] = {"256x64_w4x1", 1}, c[n++] = {"128x64_w2x2", 1}, c[n++] = {"64x64_w2x1", 1};
    } else if (a.Cout % 256 == 0 && T >= 6) {
        // 3x3: 16 waves of 64x64; 1x1: the persistent kernel (falls through when not admissible)
        // 1x1 without a residual and a very long K loop (the 2048 -> 512 conv1 of layer4): the deep-X ring.
        // A/B at batch 32 (gpurun_out/r2b): 81 -> 71 us there, but 87 -> 90 us on layer3's 1024 -> 256 -
        // those are not short of HBM requests in flight, so they keep the 2-slot form.
        const bool no_x3 = env().no_x3;      // A/B and bisecting
        // Round 6: the tuner's pick flipped on layer3's 1024 -> 256 conv1 (22 launches; in-place identity blocks and the loader /
        // consumer 3x3 have changed what runs around it since round 2): 88-90 -> 82-84 us per launch, +0.9 % on the step
        // (gpurun_out/r6tune32).  256-channel outputs take the deep-X ring from K = 1024; DIRTORCH_AMD_X3_K2048 = the old rule.
        const bool x3 = a.R * a.S == 1 && !a.res && !no_x3 && (T >= 32 || (T >= 16 && a.Cout <= 256 && !env().x3_k2048));
        // 3x3 over 256 / 512 channels: the plane-at-a-time patch kernel (conv_patch.hip) - falls through to the
        // 16-wave implicit-GEMM tile where it is not admissible (stride 2, odd widths) or too few tiles
        const bool no_ps = env().no_patchs;  // A/B and bisecting
        if (a.R * a.S > 1 && !no_ps) c[n++] = {"256x256_patch3x3s", 1};
        // (conv_ring.hip's 128x256_ring1x1 - split loader / consumer waves - ties the persistent kernel on these layers
        // inside the network, gpurun_out/r3f-r3h: it stays a tuner candidate; an experiments build + DIRTORCH_AMD_EXPERIMENTS=1 puts it first, for A/B)
#ifdef DIR_EXPERIMENTS
        if (a.R * a.S == 1 && !a.res && env().experiments) c[n++] = {"128x256_ring1x1", 1};
#endif
        if (a.R * a.S == 1 && !a.res && env().lc1x1) c[n++] = {"256x256_lc1x1", 1};   // (A/B: DIRTORCH_AMD_LC1X1)
        c[n++] = {a.R * a.S > 1 ? "256x256_w4x4" : (x3 ? "256x256_persist1x1_x3" : "256x256_persist1x1"), 1};
        c[n++] = {"256x256_w4x2", 1}, c[n++] = {"128x128_w2x2", 1};
        // small M (batch 1 at the deep stages): the 4-slot ring hides the fill latency of a long K
        // loop; with fewer than ~100 tiles even that leaves CUs idle and split-K takes over
        c[n++] = {"64x128_w2x2_s4", 1}, c[n++] = {"64x128_w2x2", 1};
    } else if (a.Cout % 256 == 0 && T >= 3) {
        c[n++] = {"128x256_w2x4_s3_k32", 2}, c[n++] = {"256x256_w4x2", 1}, c[n++] = {"128x128_w2x2", 1};
        c[n++] = {"128x64_w2x2", 1}, c[n++] = {"64x64_w2x1", 1};   // short K: more, smaller tiles
    } else {
        // strided 3x3 (the first conv2 of layer2): the gather touches every other pixel, so the 128-B K rows of
        // the BK = 64 tile halve the number of requests per byte (A/B gpurun_out/r2t: 273 -> 224 us)
        if (a.R * a.S > 1 && a.stride > 1) c[n++] = {"256x128_w4x2_s3", 1};
        c[n++] = {"256x128_w4x2_s3_k32", 2}, c[n++] = {"128x128_w2x2", 1};
        c[n++] = {T >= 8 ? "64x128_w2x2_s4" : "64x128_w2x2", 1}, c[n++] = {"64x64_w2x1", 1};
    }
    // The small-map regime (round 6, distilled from the tuner at batch 1 / batch 4 of 1024^2 and ResNet-50 at 64 x 224^2,
    // gpurun_out/r6small): when a long-K layer with 256+ outputs cannot fill the chip with 256-wide tiles and 128 x 128 tiles give
    // it less than two rounds, 64 x 128 tiles (48 KB: two workgroups per CU, one's fill under the other's MFMAs) are 8-10 %
    // faster (same box: +0.5 % on the whole step at batch 4 and on config A); below ~250 of those, 64 x 64 tiles on a 4-slot ring
    // (one per CU at 4 096 pixels x 256 channels, 64 KB so that a second stream's workgroup fits beside it).
    if (!env().no_smallmap && a.Cout % 256 == 0 && T >= 6 && !(a.R * a.S > 1 && a.Cin >= 128 && !env().no_patchw &&
                                         conv_patch3x3w_admissible(a) && (long)ceil_div(a.M, 512) * (a.Cout / 128) >= 192)) {
        const long t256 = (long)ceil_div(a.M, 256) * (a.Cout / 256), t128 = (long)ceil_div(a.M, 128) * (a.Cout / 128);
        const long t64x128 = (long)ceil_div(a.M, 64) * (a.Cout / 128), t64 = (long)ceil_div(a.M, 64) * (a.Cout / 64);
        if (t256 < 192 && t128 < 512) {
            const int v1 = find_variant("64x128_w2x2"), v2 = find_variant("64x64_w2x2_s4"), v3 = find_variant("64x64_small_s4k2");
            const bool ok2 = v2 >= 0 && conv_variant_admissible(v2, a), ok3 = v3 >= 0 && conv_variant_admissible(v3, a);
            // Long K loops in this regime (the tuner at batch 4 of 1024^2 and ResNet-50 at 64 x 224^2, scripts/exp_tune_any.py):
            //   * layer4's 3x3 (K = 4608) with 96-191 tiles of 128 x 128: that tile with split-K (4 096 pixels x 512 channels: 62 -> 38 us
            //     per launch against the 64 x 128 tile, 3 136 pixels: 38 -> 33) - the small tiles stream the 4.7 MB filter once per 64 pixels;
            //   * layer4's 2048 -> 512 conv1 (32 K-steps): the 64 x 128 tile on its four-slot ring (33 -> 21 us), not the two-slot one.
            if (a.R * a.S > 1 && T >= 64 && t128 >= 96 && t128 < 192) {
                const int v4 = find_variant("128x128_w2x2");
                if (v4 >= 0 && conv_variant_admissible(v4, a)) return v4;
            }
            if (a.R * a.S == 1 && t64x128 >= 256 && t64x128 < 384 && T >= 32) {   // (one workgroup per CU: no second one to hide the fill)
                const int v5 = find_variant("64x128_w2x2_s4");
                if (v5 >= 0 && conv_variant_admissible(v5, a)) return v5;
            }
            //   * 512+ output channels with 1.25+ rounds of 128 x 128 tiles (config A's layer4 1x1 convs, 392-400 tiles: 20-32 -> 17-24 us): the 64-pixel
            //     tile streams the wider weight matrix twice as often - those keep the list's 128 x 128 tile.
            const bool wide_n = a.Cout >= 512 && t128 >= 320;
            if (!wide_n && t64x128 >= 256 && v1 >= 0 && conv_variant_admissible(v1, a)) return v1;
            if (t64x128 < 192 && t64 >= 192) {
                // (conv_small.hip's two-K-steps-per-stage tile is 8 % faster here on ONE stream - batch 1 at 1024^2 777 -> 839 img/s -
                // and, at 128 KB of LDS, 3-4 % slower on two to four: DIRTORCH_AMD_SMALL_K2 for callers that do not overlap forwards)
                if (env().small_k2 && ok3) return v3;
                if (ok2) return v2;
            }
            // Below 192 tiles of 64 x 64 the list further down fell to split-K on 64 x 128 tiles - a cliff native-size images sit right
            // under (683 x 1024: 43 x 64 pixels in layer3 = 172 tiles; the tuner, scripts/exp_batch1_tune.py: 16 -> 10 us per conv1).
            // conv_small.hip with two K-steps per stage instead, down to 32 tiles (layer4's 2048 -> 512 conv1 at 1024^2: 19 -> 14 us);
            // gpurun_out/r6b1rules, img/s on 1 / 2 / 4 streams: 683 x 1024 760 / 1128 / 1097 -> 960 / 1330 / 1287, 500 x 375 1010 /
            // 1590 / 1550 -> 1137 / 2100 / 2094, nothing lost at 1024^2 or 768 x 1024.  (K >= 4096, layer4's 3x3, keeps split-K.)
            if (t64x128 < 192 && t64 >= 32 && t64 < 192 && T < 64) {
                if (ok3) return v3;
                if (ok2) return v2;
            }
        }
    }
    // Ragged maps (round 6, distilled from the tuner on configs[4]'s three scales, scripts/exp_multiscale_tune.py): the 16 x 32-pixel
    // patch tile wastes what a map leaves of its last tiles (107^2: 20 %, 54^2: 29 %, 38^2: 53 %) where the implicit-GEMM tile walks
    // the flattened pixels; both lose the idle part of their last round of workgroups.  With the 16-wave tile at 0.91 of the patch
    // kernel's rate on full tiles (133 vs 121 us on layer3 at batch 32): 16 x 107^2 x 256 channels 0.70 vs 0.85 -> the flattened tile
    // (tuner: 216 -> 201 us), 16 x 75^2 0.69 vs 0.63 and 16 x 54^2 0.71 vs 0.65 -> the patch kernel (tuner: the same), 16 x 38^2 x 512
    // 0.35 vs 0.65 -> the flattened tile.  DIRTORCH_AMD_NO_SMALLMAP switches this off with the other distilled rules.
    const int v_pw = find_variant("512x128_patch3x3w");
    long pw_wgs = 0;   // (set when the map fills at least 60 % of its tiles: a 7 x 7 map's one tile per image is no workgroup to count)
    if (v_pw >= 0 && a.R * a.S > 1 && a.Cin >= 128 && !env().no_patchw && conv_variant_admissible(v_pw, a)) {
        pw_wgs = (long)a.B * ceil_div(a.OH, 16) * ceil_div(a.OW, 32) * (a.Cout / 128);
        const double fill = (double)a.M / ((double)(pw_wgs / (a.Cout / 128)) * 512.0);
        const int v_g = find_variant("256x256_w4x4");
        if (!env().no_smallmap && a.Cout % 256 == 0 && v_g >= 0 && conv_variant_admissible(v_g, a)) {
            auto round_eff = [](long wgs) { return (double)wgs / (double)(ceil_div((int)wgs, 256) * 256L); };
            const long t256 = (long)ceil_div(a.M, 256) * (a.Cout / 256);
            const double eff_p = fill * round_eff(pw_wgs);
            const double eff_g = 0.91 * round_eff(t256);
            if (t256 >= 176 && eff_g > 1.05 * eff_p) return v_g;
        }
        if (fill < 0.6) pw_wgs = 0;
    }
    int last = -1, prev = -1;
    for (int i = 0; i < n; ++i) {
        const int v = find_variant(c[i].name);
        if (v < 0 || !conv_variant_admissible(v, a)) continue;
        prev = last;
        last = v;
        // (the patch kernel's workgroups are per-image tiles: on a ragged map more than ceil(M / 512))
        const long tiles = v == v_pw && pw_wgs && !env().no_smallmap ? pw_wgs : (long)ceil_div(a.M, kVariants[v].BM) * (a.Cout / kVariants[v].BN);
        if (tiles >= 192L * c[i].wg_per_cu) return v;
    }
    // nothing fills the chip: the smallest tile, unless it is the split-K fallback of a list whose
    // deep-ring sibling still gets ~100 workgroups
    if (prev >= 0 && last >= 0 && kVariants[last].launch_sk[0] != nullptr &&
        strcmp(kVariants[prev].name, "64x128_w2x2_s4") == 0 &&
        (long)ceil_div(a.M, 64) * (a.Cout / 128) >= 96)
        return prev;
    return last;
}

// Two-source form (conv3 + downsample of the first block of layers 2-4): the one kernel that carries it.
// History: conv_igemm.hip's own 256x256 tile (rounds 2-3: 0.359 / 0.261 / 0.206 ms on the first blocks of layers 2 / 3 / 4 where the
// persistent ring takes 0.311 / 0.238 / 0.197, bit-identical), a 3-slot 128x256 tile (0.39 / 0.29 / 0.22) and a split loader /
// consumer ring (0.357 / 0.267 / 0.224) were measured and retired.
int conv_pick_dual_variant(const ConvArgs& a, bool any_size) {
    const bool ok = a.x2 && a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.OH && a.W == a.OW &&
                    a.Cin % 64 == 0 && a.Cin2 % 64 == 0 && a.res == nullptr && a.ksplit <= 1;
    if (!ok) return -1;
    // (any_size: the op-level entry point runs the form on small shapes too - the tests')
    // layer2's first block (K = 128 + 256): weights stationary in registers, from ~4 pixel tiles per persistent workgroup
    // (round 6, A/B at batch 32 in profiles/r06_wregd.txt; DIRTORCH_AMD_NO_WREGD: the DUAL ring again)
    {
        const int v = find_variant("64x256_wregd1x1");
        if (v >= 0 && !env().no_wregd && conv1x1_wregd_admissible(a) &&
            (any_size || (long)ceil_div(a.M, 64) * (a.Cout / 256) >= 1024))
            return v;
    }
    if (a.Cout % 256 != 0 || (!any_size && (long)ceil_div(a.M, 256) * (a.Cout / 256) < 192)) return -1;
    // (conv_persistlc.hip - the same ring with loader / consumer wave roles - measured on layers 3-4, gpurun_out/r6lc1x1*: 4-8 % off the
    // launch by per-layer events, nothing on the whole step in two same-box A/Bs, and 5-15 % SLOWER on the plain 1x1 convs inside the
    // network; an opt-in (DIRTORCH_AMD_LC1X1) and a tuner candidate, not a default)
    if (env().lc1x1) {
        const int vl = find_variant("256x256_lc1x1");
        if (vl >= 0 && conv1x1_lc_admissible(a)) return vl;
    }
    const int v = find_variant("256x256_persist1x1_x3");
    if (v < 0 || kVariants[v].launch_dual[0] == nullptr) return -1;
    return v;
}

// ---- split-K ---------------------------------------------------------------------------------------
// y = act(sum_z partial[z] + bias (+ res)): the z order is fixed, so the result does not depend on
// which workgroup finished first.  8 channels per lane.
template <class DT>
__global__ void conv_splitk_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                            const uint16_t* res, uint16_t* y,   // (no __restrict__: in-place identity blocks pass res == y)
                                            long total8, int Cout, long MC, int ksplit, int relu, int* ovf_flag) {
    const long i8 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i8 >= total8) return;
    const long o = i8 * 8;
    const int n = (int)(o % Cout);
    float v[8];
    const f32x4_t b0 = *(const DIR_GLOBAL f32x4_t*)(bias + n), b1 = *(const DIR_GLOBAL f32x4_t*)(bias + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = b0[e], v[4 + e] = b1[e];
    for (int z = 0; z < ksplit; ++z) {
        const float* p = partial + (size_t)z * MC + o;
        const f32x4_t p0 = *(const DIR_GLOBAL f32x4_t*)p, p1 = *(const DIR_GLOBAL f32x4_t*)(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += p0[e], v[4 + e] += p1[e];
    }
    if (res) {
        const u32x4_t rv = gload16(res + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float lo, hi;
            DT::unpack(rv[e], lo, hi);
            v[2 * e] += lo;
            v[2 * e + 1] += hi;
        }
    }
    if (relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    u32x4_t ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = DT::pack(v[2 * e], v[2 * e + 1]);
    gstore16(y + o, ov);
    Ovf<DT> ovf;
    ovf.see(ov);
    ovf.flush(ovf_flag);
}

size_t conv_splitk_bytes(const ConvArgs& a, int ksplit) {
    return ksplit > 1 ? (size_t)ksplit * a.M * a.Cout * sizeof(float) : 0;
}

// When a layer has too few output tiles to give every CU work (batch 1 at the deep stages), the K loop
// is cut into slices that run as separate workgroups.  Only the implicit-GEMM variants split, only
// when each slice keeps >= 4 K-steps, and only within the scratch the engine reserves.
int conv_splitk_factor(int v, const ConvArgs& a) {
    if (v < 0 || v >= kNumVariants || kVariants[v].launch_sk[0] == nullptr || a.Cin == 16) return 1;
    const ConvVariant& cv = kVariants[v];
    const long tiles = (long)ceil_div(a.M, cv.BM) * (a.Cout / cv.BN);
    const int T = a.Ktot / cv.BK;
    if (tiles > 160 || T < 8) return 1;
    int s = (int)std::min<long>(std::min<long>(8, 512 / tiles), T / 4);
    while (s > 1 && conv_splitk_bytes(a, s) > kSplitKMaxBytes) --s;
    return s < 2 ? 1 : s;
}

int conv_launch(const ConvArgs& a, int dtype, int variant, hipStream_t stream) {
    if (a.Cout % 64 != 0) return fail(DIR_ERR_INVALID, "conv: Cout must be a multiple of 64");
    const bool cin16 = (a.Cin == 16);
    if (cin16) {
        if (a.R != 4 || a.S != 4 || a.stride != 1)
            return fail(DIR_ERR_INVALID, "conv: Cin == 16 is only the 4x4 s1 space-to-depth stem");
    } else if (a.Cin % 64 != 0) {
        return fail(DIR_ERR_INVALID, "conv: Cin must be a multiple of 64 (or 16 for the stem)");
    }
    if (a.R > 4 || a.S > 4) return fail(DIR_ERR_INVALID, "conv: filter larger than 4x4");
    if ((long)a.B * a.H * a.W * a.Cin >= (1L << 30) || (long)a.M * a.Cout >= (1L << 30) ||
        (long)a.Cout * a.Ktot >= (1L << 30))
        return fail(DIR_ERR_INVALID, "conv: tensor exceeds 2^31 bytes; lower the batch");
    if (((uintptr_t)a.x & 15) || ((uintptr_t)a.w & 15) || ((uintptr_t)a.y & 15) ||
        ((uintptr_t)a.res & 15) || ((uintptr_t)a.bias & 15))
        return fail(DIR_ERR_INVALID, "conv: tensors must be 16-byte aligned");
    if (dtype != DIR_BF16 && dtype != DIR_FP16) return fail(DIR_ERR_INVALID, "conv: bad dtype");
    if (a.x2) {   // two-source K: conv3 + downsample in one GEMM
        if (variant < 0) variant = conv_pick_dual_variant(a);
        if (variant < 0 || variant >= kNumVariants || kVariants[variant].launch_dual[0] == nullptr ||
            a.Cout % kVariants[variant].BN != 0 || (kVariants[variant].kind == 9 && !conv1x1_wregd_admissible(a)) || (kVariants[variant].kind == 10 && !conv1x1_lc_admissible(a)) || a.R != 1 || a.S != 1 || a.stride != 1 || a.res || a.ksplit > 1 ||
            a.Cin2 % 64 != 0 || a.Ktot != a.Cin + a.Cin2 || ((uintptr_t)a.x2 & 15) ||
            (long)a.B * a.H2 * a.W2 * a.Cin2 >= (1L << 30))
            return fail(DIR_ERR_INVALID, "conv: no two-source form for this shape / variant");
        hipError_t e = kVariants[variant].launch_dual[dtype](a, stream);
        if (e != hipSuccess)
            return fail(DIR_ERR_HIP, std::string("conv launch ") + kVariants[variant].name + "/dual: " + hipGetErrorString(e));
        return DIR_OK;
    }
    if (variant < 0) variant = conv_pick_variant(a);
    if (!conv_variant_admissible(variant, a))
        return fail(DIR_ERR_INVALID, "conv: variant not admissible for this shape");
    const ConvVariant& cv = kVariants[variant];
    if (a.ksplit > 1) {
        if (cv.launch_sk[0] == nullptr || cin16)
            return fail(DIR_ERR_INVALID, "conv: this variant has no split-K form");
        if (!a.partial || ((uintptr_t)a.partial & 15))
            return fail(DIR_ERR_INVALID, "conv: split-K needs a 16-byte aligned fp32 scratch buffer");
        if (a.ksplit > a.Ktot / cv.BK) return fail(DIR_ERR_INVALID, "conv: more K slices than K-steps");
    }
    hipError_t e = cv.kind == 1   ? conv_patch3x3_launch(a, dtype, stream)
                   : cv.kind == 5 ? conv_patch3x3s_launch(a, dtype, stream)
                   : cv.kind == 6 ? conv_patch3x3w_launch(a, dtype, stream)
                   : cv.kind == 2 ? conv1x1_persist_launch(a, dtype, stream)
                   : cv.kind == 4 ? conv1x1_persist_launch(a, dtype, stream, true)
                   : cv.kind == 3 ? conv1x1_wreg_launch(a, dtype, stream)
#ifdef DIR_EXPERIMENTS
                   : cv.kind == 7 ? conv1x1_ring_launch(a, dtype, stream)
#endif
                   : cv.kind == 8 ? conv_patch64_lc_launch(a, dtype, stream)
                   : cv.kind == 12 ? conv_patch3x3s2_launch(a, dtype, stream)
                   : cv.kind == 10 ? conv1x1_lc_launch(a, dtype, stream)
                   : cv.kind == 11 ? conv_small_launch(a, dtype, cv.stages, stream)
                   : a.ksplit > 1 ? cv.launch_sk[dtype](a, stream)
                                  : (cin16 ? cv.launch16 : cv.launch)[dtype](a, stream);
    if (e != hipSuccess)
        return fail(DIR_ERR_HIP, std::string("conv launch ") + cv.name + ": " + hipGetErrorString(e));
    if (a.ksplit > 1) {
        const long MC = (long)a.M * a.Cout, total8 = MC / 8;
        const dim3 grid((unsigned)((total8 + 255) / 256));
        if (dtype == DIR_BF16)
            hipLaunchKernelGGL(conv_splitk_finalize_kernel<BF16>, grid, dim3(256), 0, stream, a.partial, a.bias,
                               a.res, a.y, total8, a.Cout, MC, a.ksplit, a.relu, a.ovf);
        else
            hipLaunchKernelGGL(conv_splitk_finalize_kernel<FP16>, grid, dim3(256), 0, stream, a.partial, a.bias,
                               a.res, a.y, total8, a.Cout, MC, a.ksplit, a.relu, a.ovf);
        e = hipGetLastError();
        if (e != hipSuccess) return fail(DIR_ERR_HIP, std::string("conv split-K finalize: ") + hipGetErrorString(e));
    }
    return DIR_OK;
}

// ---- naive checker kernel ------------------------------------------------------------------------
template <class DT>
__global__ void conv_naive_kernel(const ConvArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)a.M * a.Cout) return;
    const int n = (int)(idx % a.Cout);
    const int m = (int)(idx / a.Cout);
    const int OHW = a.OH * a.OW;
    const int b = m / OHW, rem = m - b * OHW, oh = rem / a.OW, ow = rem - oh * a.OW;
    float acc = 0.f;
    for (int r = 0; r < a.R; ++r) {
        const int ih = oh * a.stride - a.pad + r;
        if ((unsigned)ih >= (unsigned)a.H) continue;
        for (int s = 0; s < a.S; ++s) {
            const int iw = ow * a.stride - a.pad + s;
            if ((unsigned)iw >= (unsigned)a.W) continue;
            const uint16_t* xp = a.x + ((size_t)(b * a.H + ih) * a.W + iw) * a.Cin;
            const uint16_t* wp = a.w + (size_t)n * a.Ktot + (r * a.S + s) * a.Cin;
            for (int c = 0; c < a.Cin; ++c) acc = fmaf(DT::to_f32(xp[c]), DT::to_f32(wp[c]), acc);
        }
    }
    float v = acc + a.bias[n];
    if (a.res) v += DT::to_f32(a.res[(size_t)m * a.Cout + n]);
    if (a.relu) v = fmaxf(v, 0.f);
    a.y[(size_t)m * a.Cout + n] = DT::from_f32(v);
}

int conv_launch_naive(const ConvArgs& a, int dtype, hipStream_t stream) {
    const long total = (long)a.M * a.Cout;
    const int threads = 256;
    const long blocks = (total + threads - 1) / threads;
    if (blocks >= (1L << 31)) return fail(DIR_ERR_INVALID, "naive conv: too many blocks");
    if (dtype == DIR_BF16)
        hipLaunchKernelGGL(conv_naive_kernel<BF16>, dim3((unsigned)blocks), dim3(threads), 0, stream, a);
    else
        hipLaunchKernelGGL(conv_naive_kernel<FP16>, dim3((unsigned)blocks), dim3(threads), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DIR_ERR_HIP, std::string("naive conv: ") + hipGetErrorString(e));
    return DIR_OK;
}

}  // namespace dir
