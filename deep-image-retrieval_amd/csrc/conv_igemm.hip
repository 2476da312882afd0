// conv_igemm.hip — implicit-GEMM convolution for gfx950 (MFMA 32x32x16, 16-bit in / fp32 acc).
//
// Replaces every Conv2d + BatchNorm2d(eval) (+ residual add) (+ ReLU) of the reference trunk
// (dirtorch/nets/backbones/resnet.py:56-63,67-87,115-118,136-141) with one fused kernel family.
//
// GEMM view (per launch):   Y[m][n] = act( sum_k X[m][k] * Wt[n][k] + bias[n] (+ res[m][n]) )
//   m = (b, oh, ow) output pixel, n = output channel, k = (r, s, c) filter tap x input channel.
//   X is never materialised: for K-step (r, s, c0..c0+63) the 64-channel slice of input pixel
//   (oh*stride + r - pad, ow*stride + s - pad) is one contiguous 128-byte run of the NHWC tensor.
//   Out-of-image taps read a 16-byte device zero page instead (zero padding).
//   The stem (7x7 s2, Cin = 3) arrives as a 4x4 s1 convolution over the 2x2 space-to-depth image
//   (Cin = 16): one K-step = one filter row = four neighbouring pixels = the same 128-byte run.
//
// Tile: BM output pixels x BN output channels per workgroup, K-step 64.  LDS holds two stages of
//   X-tile [BM][64] + W-tile [BN][64] 16-bit, rows of 128 B, 16-byte chunks XOR-swizzled with
//   ((row >> 1) & 7) so the MFMA fragment reads (ds_read_b128: row = lane & 31, chunk = 2*ks + lane/32)
//   are bank-conflict free.  Staging is either LDS-DMA (global_load_lds_dwordx4; destination is
//   lane-linear, so the swizzle is applied to the per-lane SOURCE chunk) or through registers.
// MFMA operand roles are swapped (A = weights, B = pixels) so each lane ends up holding 4
//   consecutive output channels of one pixel; the epilogue stages the wave's accumulators through
//   LDS as fp32 and writes whole 16-byte runs (8 channels) with bias / residual / ReLU fused.
#include "dir_common.h"
#include "conv_igemm.h"

namespace dir {

template <class DT, int BM, int BN, int WGM, int WGN, int STG, bool CIN16>
__global__ void __launch_bounds__(64 * WGM * WGN) conv_igemm_kernel(const ConvArgs a) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int TM = BM / WGM / 32;  // pixel (B-operand) tiles per wave
    constexpr int TN = BN / WGN / 32;  // channel (A-operand) tiles per wave
    constexpr int NA = BM * 8 / NT;    // 16-byte X chunks per thread per stage
    constexpr int NB = BN * 8 / NT;    // 16-byte W chunks per thread per stage
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "chunk split");
    static_assert((NT / 8) % 16 == 0, "swizzle term must be constant per thread");
    constexpr int XS = BM * 128;             // bytes of the X tile of one stage
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int EROW = TN * 128 + 16;      // epilogue: one pixel row of TN*32 fp32 + pad
    typedef typename DT::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n;  // n fastest: blocks sharing an X tile run on one XCD
    const int tile_m = wg / a.tiles_n;

    // ---- per-thread source bookkeeping (constant over the K loop) -------------------------------
    const int slot = tid & 7;
    const int srcchunk = slot ^ ((tid >> 4) & 7);
    int xoff[NA];
    uint32_t xmask[NA];
    const int OHW = a.OH * a.OW;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = i * (NT / 8) + (tid >> 3);
        const int m = tile_m * BM + row;
        const bool mvalid = m < a.M;
        const int mm = mvalid ? m : 0;
        const int b = mm / OHW;
        const int rem = mm - b * OHW;
        const int oh = rem / a.OW;
        const int ow = rem - oh * a.OW;
        const int ih0 = oh * a.stride - a.pad;
        const int iw0 = ow * a.stride - a.pad;
        xoff[i] = ((b * a.H + ih0) * a.W + iw0) * a.Cin + srcchunk * 8;
        uint32_t mask = 0;
        if (CIN16) {
            const int sp = srcchunk >> 1;  // which of the 4 pixels of the filter row this chunk is
            const bool wok = (unsigned)(iw0 + sp) < (unsigned)a.W;
            for (int r = 0; r < a.R; ++r)
                if (mvalid && wok && (unsigned)(ih0 + r) < (unsigned)a.H) mask |= 1u << r;
        } else {
            for (int r = 0; r < a.R; ++r)
                for (int s = 0; s < a.S; ++s)
                    if (mvalid && (unsigned)(ih0 + r) < (unsigned)a.H &&
                        (unsigned)(iw0 + s) < (unsigned)a.W)
                        mask |= 1u << (r * a.S + s);
        }
        xmask[i] = mask;
    }
    int woff[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = i * (NT / 8) + (tid >> 3);
        woff[i] = (tile_n * BN + row) * a.Ktot + srcchunk * 8;
    }

    // ---- staging ------------------------------------------------------------------------------
    u32x4_t xr[STG == STG_REG ? NA : 1];
    u32x4_t wr[STG == STG_REG ? NB : 1];
    (void)xr;
    (void)wr;

    // K-step t covers filter tap index `tap` (= r*S+s, or r for the stem) and channels c0..c0+63.
    auto issue = [&](int t, int tap, int koff, char* stage) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool ok = (xmask[i] >> tap) & 1u;
            const uint16_t* src = ok ? a.x + (xoff[i] + koff) : a.zero;
            if (STG == STG_GLDS) {
                __builtin_amdgcn_global_load_lds((const DIR_GLOBAL void*)src,
                                                 (DIR_LDS void*)(stage + (i * NT + wave * 64) * 16),
                                                 16, 0, 0);
            } else {
                xr[STG == STG_REG ? i : 0] = gload16(src);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const uint16_t* src = a.w + (woff[i] + t * 64);
            if (STG == STG_GLDS) {
                __builtin_amdgcn_global_load_lds(
                    (const DIR_GLOBAL void*)src,
                    (DIR_LDS void*)(stage + XS + (i * NT + wave * 64) * 16), 16, 0, 0);
            } else {
                wr[STG == STG_REG ? i : 0] = gload16(src);
            }
        }
    };
    auto commit = [&](char* stage) {  // register staging only: registers -> LDS
        if (STG == STG_REG) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                *(u32x4_t*)(stage + (i * NT + tid) * 16) = xr[STG == STG_REG ? i : 0];
#pragma unroll
            for (int i = 0; i < NB; ++i)
                *(u32x4_t*)(stage + XS + (i * NT + tid) * 16) = wr[STG == STG_REG ? i : 0];
        }
    };

    // ---- fragment read offsets -----------------------------------------------------------------
    const int wn = wave % WGN;
    const int wm = wave / WGN;
    const int lrow = lane & 31;
    const int lhi = lane >> 5;
    const int lswz = (lane >> 1) & 7;
    int loff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) loff[ks] = lrow * 128 + (((2 * ks + lhi) ^ lswz) << 4);
    const int xbase = (wm * TM * 32) * 128;
    const int wbase = XS + (wn * TN * 32) * 128;

    f32x16_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](const char* stage) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            frag_t wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i)
                wf[i] = *(const frag_t*)(stage + wbase + i * 4096 + loff[ks]);
#pragma unroll
            for (int j = 0; j < TM; ++j)
                xf[j] = *(const frag_t*)(stage + xbase + j * 4096 + loff[ks]);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = DT::mfma32(wf[i], xf[j], acc[i][j]);
        }
    };

    // ---- epilogue operands fetched up front: their HBM latency hides under the whole K loop ----
    constexpr int LPR = TN * 4;        // lanes covering one pixel row (8 channels each)
    constexpr int RPP = 64 / LPR;      // pixel rows per pass
    constexpr int NPASS = 32 / RPP;    // passes per 32-pixel strip
    constexpr bool PRE_RES = (TM * NPASS <= 8);  // 16 B per lane each: at most 32 VGPRs
    const int ecol = (lane % LPR) * 8;
    const int erow = lane / LPR;
    const int n_glob = tile_n * BN + wn * TN * 32 + ecol;
    const int m_epi = tile_m * BM + wm * TM * 32;
    float bias8[8];
    {
        const f32x4_t b0 = *(const DIR_GLOBAL f32x4_t*)(a.bias + n_glob);
        const f32x4_t b1 = *(const DIR_GLOBAL f32x4_t*)(a.bias + n_glob + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bias8[e] = b0[e];
            bias8[4 + e] = b1[e];
        }
    }
    u32x4_t rres[PRE_RES ? TM : 1][PRE_RES ? NPASS : 1];
    if (PRE_RES && a.res) {
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int m = m_epi + j * 32 + pass * RPP + erow;
                const int mc = m < a.M ? m : 0;   // clamped rows are never stored
                rres[PRE_RES ? j : 0][PRE_RES ? pass : 0] =
                    gload16(a.res + ((size_t)mc * a.Cout + n_glob));
            }
    }

    // ---- K loop: double-buffered, one barrier per K-step ----------------------------------------
    const int T = a.T;
    const int cpb = CIN16 ? 1 : (a.Cin >> 6);  // K-steps per filter tap
    int tap = 0, cc = 0, r = 0, s = 0;         // state of the step being ISSUED
    auto koff_now = [&]() {
        return CIN16 ? (r * a.W * 16) : ((r * a.W + s) * a.Cin + cc * 64);
    };
    auto advance = [&]() {
        if (++cc == cpb) {
            cc = 0;
            ++tap;
            if (CIN16) {
                ++r;
            } else if (++s == a.S) {
                s = 0;
                ++r;
            }
        }
    };

    char* stage0 = smem;
    char* stage1 = smem + STAGE_BYTES;
    issue(0, tap, koff_now(), stage0);
    advance();
    commit(stage0);
    if (STG == STG_GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        char* cur = (t & 1) ? stage1 : stage0;
        char* nxt = (t & 1) ? stage0 : stage1;
        const bool more = (t + 1 < T);
        if (more) {
            issue(t + 1, tap, koff_now(), nxt);
            advance();
        }
        compute(cur);
        if (more) {
            commit(nxt);
            if (STG == STG_GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }

    // ---- epilogue: acc -> LDS (fp32, pixel-major) -> bias/residual/ReLU -> 16-byte stores -------
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
            if (m < a.M) {
                float v[8] = {f0[0] + bias8[0], f0[1] + bias8[1], f0[2] + bias8[2],
                              f0[3] + bias8[3], f1[0] + bias8[4], f1[1] + bias8[5],
                              f1[2] + bias8[6], f1[3] + bias8[7]};
                const size_t o = (size_t)m * a.Cout + n_glob;
                if (a.res) {
                    const u32x4_t rv = PRE_RES ? rres[PRE_RES ? j : 0][PRE_RES ? pass : 0]
                                               : gload16(a.res + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo, hi;
                        unpack2<DT>(rv[e], lo, hi);
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
                for (int e = 0; e < 4; ++e) ov[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                gstore16(a.y + o, ov);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---- variant table ----------------------------------------------------------------------------
template <class DT, int BM, int BN, int WGM, int WGN, int STG, bool CIN16>
static hipError_t launch_variant(const ConvArgs& a, hipStream_t stream) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int TN = BN / WGN / 32;
    constexpr int EROW = TN * 128 + 16;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int EPI_BYTES = (NT / 64) * 32 * EROW;
    constexpr int LDS = (2 * STAGE_BYTES > EPI_BYTES) ? 2 * STAGE_BYTES : EPI_BYTES;
    auto kern = conv_igemm_kernel<DT, BM, BN, WGM, WGN, STG, CIN16>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    ConvArgs b = a;
    b.tiles_m = ceil_div(a.M, BM);
    b.tiles_n = a.Cout / BN;
    // a single K-step never touches the second stage: ask for half the LDS, double the residency
    const int one = (STAGE_BYTES > EPI_BYTES) ? STAGE_BYTES : EPI_BYTES;
    const int lds = a.T > 1 ? LDS : one;
    hipLaunchKernelGGL(kern, dim3(b.tiles_m * b.tiles_n), dim3(NT), lds, stream, b);
    return hipGetLastError();
}

#define DIR_VARIANT(BM, BN, WGM, WGN, STG, NAME)                                              \
    {NAME, BM, BN, 64 * WGM * WGN, STG,                                                       \
     {launch_variant<BF16, BM, BN, WGM, WGN, STG, false>,                                     \
      launch_variant<FP16, BM, BN, WGM, WGN, STG, false>},                                    \
     {nullptr, nullptr}}
// BN == 64 variants also carry the Cin == 16 (space-to-depth stem) instantiation.
#define DIR_VARIANT16(BM, BN, WGM, WGN, STG, NAME)                                            \
    {NAME, BM, BN, 64 * WGM * WGN, STG,                                                       \
     {launch_variant<BF16, BM, BN, WGM, WGN, STG, false>,                                     \
      launch_variant<FP16, BM, BN, WGM, WGN, STG, false>},                                    \
     {launch_variant<BF16, BM, BN, WGM, WGN, STG, true>,                                      \
      launch_variant<FP16, BM, BN, WGM, WGN, STG, true>}}

static const ConvVariant kVariants[] = {
    DIR_VARIANT(128, 128, 2, 2, STG_GLDS, "128x128_w2x2_glds"),
    DIR_VARIANT(128, 128, 2, 2, STG_REG, "128x128_w2x2_reg"),
    DIR_VARIANT16(128, 64, 2, 2, STG_GLDS, "128x64_w2x2_glds"),
    DIR_VARIANT16(128, 64, 2, 2, STG_REG, "128x64_w2x2_reg"),
    DIR_VARIANT16(256, 64, 4, 1, STG_GLDS, "256x64_w4x1_glds"),
    DIR_VARIANT16(256, 64, 4, 1, STG_REG, "256x64_w4x1_reg"),
    DIR_VARIANT(256, 128, 4, 2, STG_GLDS, "256x128_w4x2_glds"),
    DIR_VARIANT(256, 128, 4, 2, STG_REG, "256x128_w4x2_reg"),
    DIR_VARIANT(128, 256, 2, 4, STG_GLDS, "128x256_w2x4_glds"),
    DIR_VARIANT(256, 256, 4, 2, STG_GLDS, "256x256_w4x2_glds"),
    DIR_VARIANT(64, 128, 2, 2, STG_GLDS, "64x128_w2x2_glds"),
    DIR_VARIANT16(64, 64, 2, 1, STG_GLDS, "64x64_w2x1_glds"),
};
static constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

int conv_variant_count() { return kNumVariants; }
const ConvVariant& conv_variant(int i) { return kVariants[i]; }

bool conv_variant_admissible(int v, const ConvArgs& a) {
    if (v < 0 || v >= kNumVariants) return false;
    const ConvVariant& cv = kVariants[v];
    if (a.Cout % cv.BN != 0) return false;
    if (a.Cin == 16 && cv.launch16[0] == nullptr) return false;
    return true;
}

// Heuristic: widest channel tile that divides Cout; pixel tile as large as still leaves at least
// ~2 workgroups per CU (256 CUs) so the tail wave stays short.
int conv_pick_variant(const ConvArgs& a) {
    const int order[] = {0, 2, 6, 4, 10, 11};  // glds family, big-first fallbacks below
    (void)order;
    auto tiles = [&](int v) {
        return (long)ceil_div(a.M, kVariants[v].BM) * (a.Cout / kVariants[v].BN);
    };
    int cands[8];
    int nc = 0;
    if (a.Cout % 128 == 0) {
        cands[nc++] = 6;   // 256x128
        cands[nc++] = 0;   // 128x128
        cands[nc++] = 10;  // 64x128
    } else {
        cands[nc++] = 4;   // 256x64
        cands[nc++] = 2;   // 128x64
        cands[nc++] = 11;  // 64x64
    }
    for (int i = 0; i < nc; ++i)
        if (tiles(cands[i]) >= 512) return cands[i];
    return cands[nc - 1];
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
    if (a.R * a.S > 16) return fail(DIR_ERR_INVALID, "conv: at most 16 filter taps");
    if ((long)a.B * a.H * a.W * a.Cin >= (1L << 31) || (long)a.M * a.Cout >= (1L << 31) ||
        (long)a.Cout * a.Ktot >= (1L << 31))
        return fail(DIR_ERR_INVALID, "conv: tensor exceeds 2^31 elements; lower the batch");
    if (dtype != DIR_BF16 && dtype != DIR_FP16) return fail(DIR_ERR_INVALID, "conv: bad dtype");
    if (variant < 0) variant = conv_pick_variant(a);
    if (!conv_variant_admissible(variant, a))
        return fail(DIR_ERR_INVALID, "conv: variant not admissible for this shape");
    const ConvVariant& cv = kVariants[variant];
    hipError_t e = (cin16 ? cv.launch16 : cv.launch)[dtype](a, stream);
    if (e != hipSuccess)
        return fail(DIR_ERR_HIP, std::string("conv launch ") + cv.name + ": " + hipGetErrorString(e));
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
