// sim_split.hip — the Q x N similarity against a LARGE database on the bf16 matrix cores, at fp32 accuracy (gfx950).
//
//   scores[q][n] = sum_k queries[q][k] * database[n][k]          (dirtorch/utils/common.py:30-38, test_dir.py:150)
//
// The exact kernel (gemm_f32.hip, v_mfma_f32_32x32x2_f32) is bound by the fp32 MFMA rate: 70 queries pad to 96 rows
// and 2 * 96 * N * K FLOP at 157 TFLOP/s is 2.5 ms for the 1 006 322 x 2048 database of BASELINE config D before a
// single stall, against 1.3 ms to stream the 8.2 GB once.  Here every fp32 operand is split into three bf16 planes
//     x = h + m + l,   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)      (round to nearest even; the two
//                                                                              subtractions are exact in fp32)
// and the product is assembled from the six plane products of weight >= 2^-16,
//     x*y ~= h*h' + (h*m' + m*h') + (h*l' + l*h' + m*m'),
// each an exact fp32 number inside v_mfma_f32_32x32x16_bf16; the leading product and the five corrections are
// summed in two fp32 accumulators that meet once at the end (fewer and smaller roundings than a K-long fmaf chain).  What is dropped
// (m*l' + l*m' + l*l') is below 2^-23 |x*y| per term - under the half-ulp an fp32 product rounds by - at 6/16 of the
// fp32 MFMA time.  Same exponent range as fp32 (bf16 keeps all 8 exponent bits), so no scaling is needed and the
// kernel takes any fp32 matrices, not just unit vectors.  The result differs from the exact kernel the way a
// different summation order does (~1e-7 on unit vectors); callers that need the k-ordered fmaf chain keep
// gemm_nt_f32 (small N, or DIRTORCH_AMD_SIM_EXACT=1).
//
// PAIR form (round 4, similarity_split(..., unit_range = true): operands known to lie in (-64, 64) - L2-normalised
// descriptors): the six bf16 products make the kernel as much matrix-pipe- as HBM-bound (consumers alone 1.64 ms of the
// 2.0 at config D's sizes).  For bounded operands TWO fp16 planes do the same job:
//     x * 2^10 = h + l,   h = fp16(.), l = fp16(. - h)          (|l| <= 2^-11 |h|: ~22 bits; exact down to 2^-35 absolute)
//     x*y ~= (h*h' + (h*l' + l*h')) * 2^-20                     three products; l*l' (2^-22 relative) dropped
// on v_mfma_f32_32x32x16_f16 - half the MFMAs, two thirds of the query-plane traffic, a shorter split.  fp16 has fp32's
// precision problem in reverse (5 exponent bits): the 2^10 scale puts unit-vector entries (~2^-6) in the middle of the
// range, values of magnitude >= 64 overflow h to inf and the scores come out non-finite - loudly wrong, never silently.
//
// Work split: one 512-thread workgroup per 256 database rows, one 32-row strip per wave against all 96 query rows
// (3 accumulator blocks).  The database is used by exactly one wave, so it goes HBM -> LDS as raw fp32 by LDS-DMA
// (128-byte pieces, XOR-swizzled rows) and is split in registers right before the MFMAs; the query planes are split
// once by split_queries_kernel into an image that is byte-for-byte the LDS layout of every K slab (18 KB per 32
// k), so their re-streaming from L2 is a linear copy.  Three stages of (32 KB database + 18 KB query planes) in LDS,
// two in flight; one barrier per K slab.  Algorithmic bytes per launch: N*K*4 (database, once) + Q*N*4 (scores).
#include "dir_common.h"

namespace dir {

static constexpr int kQB = 96;                    // query rows per block (3 MFMA row blocks)
static constexpr int kPlane = kQB * 64;           // bytes of one bf16 plane of a 32-wide K slab
static constexpr int kSlabQ = 3 * kPlane;         // 18432
static constexpr int kRowsP = 256;                // database rows per workgroup
static constexpr int kSlabP = kRowsP * 128;       // 32768
static constexpr int kStage = kSlabP + kSlabQ;    // 51200
static constexpr int kStages = 3;
static constexpr int kLds = kStages * kStage;     // 153600 of 163840
// PAIR form: two fp16 planes
static constexpr int kSlabQ2 = 2 * kPlane;        // 12288
static constexpr int kStage2 = kSlabP + kSlabQ2;  // 45056
static constexpr int kLds2 = kStages * kStage2;   // 135168
static constexpr float kPairScale = 1024.f;       // 2^10: |x| < 64 stays finite in fp16, unit-vector entries mid-range

__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)lds, 16, voff, soff, 0, 0);
}

// x0, x1 -> packed bf16 planes (low half = x0)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    float a, b;
    h = BF16::pack(x0, x1);
    BF16::unpack(h, a, b);
    x0 -= a, x1 -= b;
    m = BF16::pack(x0, x1);
    BF16::unpack(m, a, b);
    x0 -= a, x1 -= b;
    l = BF16::pack(x0, x1);
}

// x0, x1 (already scaled) -> packed fp16 planes
__device__ __forceinline__ void split2h(float x0, float x1, uint32_t& h, uint32_t& l) {
    float a, b;
    h = FP16::pack(x0, x1);
    FP16::unpack(h, a, b);
    l = FP16::pack(x0 - a, x1 - b);
}

// Query planes as an image of the LDS stages: [query block][K slab of 32][plane h, m, l][96 rows][64 bytes], the
// four 16-byte chunks of a row XOR-swizzled by (row >> 2) & 3.  Rows past NQ and k past K are zero.
// PAIR: [..][plane h, l (fp16 of 2^10 x)][96 rows][64 bytes]
template <bool PAIR>
__global__ void __launch_bounds__(256) split_queries_kernel(const float* __restrict__ Q, int ldq, int NQ, int K,
                                                           uint16_t* __restrict__ img) {
    const int t = blockIdx.x, qb = blockIdx.y, T = gridDim.x;
    char* dst = (char*)img + ((size_t)qb * T + t) * (PAIR ? kSlabQ2 : kSlabQ);
    for (int item = threadIdx.x; item < kQB * 4; item += 256) {
        const int row = item >> 2, pos = item & 3;
        const int chunk = pos ^ ((row >> 2) & 3);
        const int q = qb * kQB + row;
        u32x4_t h, m, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = t * 32 + chunk * 8 + 2 * e;
            const float x0 = (q < NQ && k < K) ? Q[(size_t)q * ldq + k] : 0.f;
            const float x1 = (q < NQ && k + 1 < K) ? Q[(size_t)q * ldq + k + 1] : 0.f;
            uint32_t a, b, c = 0;
            if (PAIR)
                split2h(x0 * kPairScale, x1 * kPairScale, a, b);
            else
                split2(x0, x1, a, b, c);
            h[e] = a, m[e] = b, l[e] = c;
        }
        *(u32x4_t*)(dst + 0 * kPlane + row * 64 + pos * 16) = h;
        *(u32x4_t*)(dst + 1 * kPlane + row * 64 + pos * 16) = m;
        if (!PAIR) *(u32x4_t*)(dst + 2 * kPlane + row * 64 + pos * 16) = l;
    }
}

__global__ void __launch_bounds__(512) sim_split_kernel(const float* __restrict__ P, int ldp, int NP, int K,
                                                       const uint16_t* __restrict__ img, float* __restrict__ out,
                                                       int ldo, int NQ, int T) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int tile = blockIdx.x, qb = blockIdx.y;
    const int i0 = tile * kRowsP;
    const int rows = min(kRowsP, NP - i0);

    // the database can exceed the 4 GB a buffer descriptor spans: one descriptor per workgroup, based at its rows
    const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(P + (size_t)i0 * ldp), 0, (int)((((size_t)rows - 1) * ldp + K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_q = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)img + (size_t)qb * T * kSlabQ), 0, T * kSlabQ, 0x00020000);

    // database: instruction j = i * 8 + wave covers rows 8j .. 8j+7 (8 lanes x 16 bytes = one 128-byte piece)
    uint32_t pvoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        pvoff[i] = (uint32_t)row * (uint32_t)ldp * 4u + (uint32_t)chunk * 16u;   // rows past `rows`: out of range -> 0
    }
    const uint32_t qvoff = (uint32_t)(wave * (kSlabQ / 8) + lane * 16);          // 2304 bytes per wave: 2 + 1/4 instr.

    auto issue = [&](int u, char* stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16s(rsrc_p, stage + (i * 8 + wave) * 1024, pvoff[i], (uint32_t)u * 128u);
        char* qd = stage + kSlabP + wave * (kSlabQ / 8);
        dma16s(rsrc_q, qd, qvoff, (uint32_t)u * kSlabQ);
        dma16s(rsrc_q, qd + 1024, qvoff + 1024, (uint32_t)u * kSlabQ);
        if (lane < 16) dma16s(rsrc_q, qd + 2048, qvoff + 2048, (uint32_t)u * kSlabQ);
    };
    constexpr int kOps = 7;   // LDS-DMA instructions per wave per stage

    f32x16_t acc[3], lo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f, lo[j][e] = 0.f;

    const int boff = (wave * 32 + lrow) * 128, bswz = (lrow >> 1) & 7;
    const int aoff = kSlabP + lrow * 64, aswz = (lrow >> 2) & 3;

    auto compute = [&](const char* stage) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c0 = s * 4 + lhi * 2;
            const f32x4_t b0 = *(const f32x4_t*)(stage + boff + ((c0 ^ bswz) << 4));
            const f32x4_t b1 = *(const f32x4_t*)(stage + boff + (((c0 + 1) ^ bswz) << 4));
            u32x4_t ah[3], am[3], al[3];
            const int ach = ((s * 2 + lhi) ^ aswz) << 4;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                ah[j] = *(const u32x4_t*)(stage + aoff + 0 * kPlane + j * 2048 + ach);
                am[j] = *(const u32x4_t*)(stage + aoff + 1 * kPlane + j * 2048 + ach);
                al[j] = *(const u32x4_t*)(stage + aoff + 2 * kPlane + j * 2048 + ach);
            }
            u32x4_t bh, bm, bl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t h, m, l;
                split2(e < 2 ? b0[2 * e] : b1[2 * e - 4], e < 2 ? b0[2 * e + 1] : b1[2 * e - 3], h, m, l);
                bh[e] = h, bm[e] = m, bl[e] = l;
            }
#define DIR_MM(ACC, A, B)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) ACC[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(         \
        __builtin_bit_cast(bf16x8_t, A[j]), __builtin_bit_cast(bf16x8_t, B), ACC[j], 0, 0, 0)
            // the five correction products (weight <= 2^-8) have their own accumulator: the leading sum then sees
            // K/16 roundings instead of 6K/16, and the corrections round at 2^-8 of its scale
            DIR_MM(lo, al, bh);
            DIR_MM(lo, ah, bl);
            DIR_MM(lo, am, bm);
            DIR_MM(lo, am, bh);
            DIR_MM(lo, ah, bm);
            DIR_MM(acc, ah, bh);
#undef DIR_MM
        }
    };

    // every workgroup walks K from its own starting slab: with an 8 KiB row pitch workgroups in lock-step would
    // all be on the same few HBM channels at any instant (same rotation as gemm_nt_f32_kernel; the sum is the same)
    const int rot = (int)(((unsigned)tile * 7u) % (unsigned)T);
    auto slab = [&](int t) {
        int u = t + rot;
        return u >= T ? u - T : u;
    };
    issue(slab(0), smem);
    if (T > 1) issue(slab(1), smem + kStage);
    int cur = 0;
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T)
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(kOps) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_barrier();   // slab t landed everywhere; everyone is past slab t-1, whose slot is free
        int nxt = cur + 2;
        if (nxt >= kStages) nxt -= kStages;
        if (t + 2 < T) issue(slab(t + 2), smem + nxt * kStage);
        compute(smem + cur * kStage);
        cur = cur + 1 == kStages ? 0 : cur + 1;
    }

    // D[i][j]: lane holds database row j = lane & 31 of the strip, query rows i = 8g + 4(lane >> 5) + e
    const int n = i0 + wave * 32 + lrow;
    if (n < NP) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = qb * kQB + j * 32 + 8 * g + 4 * lhi + e;
                    if (q < NQ) out[(size_t)q * ldo + n] = acc[j][4 * g + e] + lo[j][4 * g + e];
                }
    }
}

// Timing-only experiment builds (scripts/exp_abl.sh sim_split DIR_SIM_ABL <bits>): -DDIR_SIM_ABL=<bits> - 1 = no database DMA, 2 = no query DMA,
// 4 = consumers only take the barriers.  Results are NOT valid scores.
#ifndef DIR_SIM_ABL
#define DIR_SIM_ABL 0
#endif

// ---- loader / consumer form (default) -----------------------------------------------------------------------------
// Same tiles, stages, fragments and arithmetic as sim_split_kernel above (bit-identical scores), with the work split by
// wave role: on a memory-bound CU the request queue is full, every LDS-DMA instruction holds its wave at issue until the
// queue drains, and in the kernel above the eight waves that issue the next slab are the eight waves that should be
// splitting and multiplying this one - the two phases serialise (the same finding as conv_ring.hip,
// profiles/r03_ring_ablation.txt).  Here a workgroup has TWELVE waves (148 VGPRs: three waves per SIMD fit): waves 0-7
// are the consumers of the old kernel (one 32-row strip each, no memory ops until the final score store), waves 8-11
// only issue LDS-DMA - 8 database pieces + 4 or 5 KB of the query image per slab each - and wait for it.  One
// s_barrier per K slab is the hand-off both ways (slab t landed / slot of slab t - 1 is free).
template <bool PAIR>
__global__ void __launch_bounds__(768) sim_split_lc_kernel(const float* __restrict__ P, int ldp, int NP, int K,
                                                          const uint16_t* __restrict__ img, float* __restrict__ out,
                                                          int ldo, int NQ, int T) {
    constexpr int kSlabQ = PAIR ? dir::kSlabQ2 : dir::kSlabQ;     // (shadow the three-plane constants)
    constexpr int kStage = PAIR ? dir::kStage2 : dir::kStage;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, qb = blockIdx.y;
    const int i0 = tile * kRowsP;
    const int rows = min(kRowsP, NP - i0);
    // every workgroup walks K from its own starting slab (see sim_split_kernel)
    const int rot = (int)(((unsigned)tile * 7u) % (unsigned)T);

    if (wave >= 8) {
        // ================================ loaders ==============================================================
        const int lw = wave - 8;
        const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(P + (size_t)i0 * ldp), 0, (int)((((size_t)rows - 1) * ldp + K) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_q = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const char*)img + (size_t)qb * T * kSlabQ), 0, T * kSlabQ, 0x00020000);
        // database piece j = lw * 8 + i covers rows 8j .. 8j+7 (8 lanes x 16 bytes = one 128-byte run per row)
        uint32_t pvoff[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (lw * 8 + i) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            pvoff[i] = (uint32_t)row * (uint32_t)ldp * 4u + (uint32_t)chunk * 16u;   // rows past `rows`: out of range -> 0
        }
        // query image of a slab: 18 pieces of 1 KiB; loaders 0 / 1 take 5, loaders 2 / 3 take 4 (PAIR: 12 pieces, 3 each)
        const int q0 = PAIR ? lw * 3 : (lw < 2 ? lw * 5 : 10 + (lw - 2) * 4);
        const bool five = !PAIR && lw < 2;
        auto issue = [&](int t) __attribute__((always_inline)) {
            int u = t + rot;
            u = u >= T ? u - T : u;
            char* stage = smem + (t % kStages) * kStage;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (!(DIR_SIM_ABL & 1)) dma16s(rsrc_p, stage + (lw * 8 + i) * 1024, pvoff[i], (uint32_t)u * 128u);
#pragma unroll
            for (int i = 0; i < (PAIR ? 3 : 4); ++i)
                if (!(DIR_SIM_ABL & 2))
                    dma16s(rsrc_q, stage + kSlabP + (q0 + i) * 1024, (uint32_t)((q0 + i) * 1024 + lane * 16), (uint32_t)u * kSlabQ);
            if (five && !(DIR_SIM_ABL & 2)) dma16s(rsrc_q, stage + kSlabP + (q0 + 4) * 1024, (uint32_t)((q0 + 4) * 1024 + lane * 16), (uint32_t)u * kSlabQ);
        };
        issue(0);
        if (T > 1) issue(1);
        for (int t = 0; t < T; ++t) {
            // this wave's part of slab t has landed; its newest 12 / 13 ops (slab t + 1) may stay in flight
            if (t + 1 < T) {
                if (PAIR) {
                    asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
                } else if (five) {
                    asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            ring_barrier();   // hand-off t: slab t is complete; the consumers have left slab t - 1
            if (t + 2 < T) issue(t + 2);
        }
        return;
    }

    // ==================================== consumers =============================================================
    const int lrow = lane & 31, lhi = lane >> 5;
    f32x16_t acc[3], lo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f, lo[j][e] = 0.f;
    const int boff = (wave * 32 + lrow) * 128, bswz = (lrow >> 1) & 7;
    const int aoff = kSlabP + lrow * 64, aswz = (lrow >> 2) & 3;

    int cur = 0;
    for (int t = 0; t < T; ++t) {
        ring_barrier();   // hand-off t (see the loaders)
        const char* stage = smem + cur * kStage;
#pragma unroll
        for (int s = 0; s < ((DIR_SIM_ABL & 4) ? 0 : 2); ++s) {
            const int c0 = s * 4 + lhi * 2;
            const f32x4_t b0 = *(const f32x4_t*)(stage + boff + ((c0 ^ bswz) << 4));
            const f32x4_t b1 = *(const f32x4_t*)(stage + boff + (((c0 + 1) ^ bswz) << 4));
            u32x4_t ah[3], am[3], al[3];
            const int ach = ((s * 2 + lhi) ^ aswz) << 4;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                ah[j] = *(const u32x4_t*)(stage + aoff + 0 * kPlane + j * 2048 + ach);
                am[j] = *(const u32x4_t*)(stage + aoff + 1 * kPlane + j * 2048 + ach);
                if (!PAIR) al[j] = *(const u32x4_t*)(stage + aoff + 2 * kPlane + j * 2048 + ach);
            }
            u32x4_t bh, bm, bl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t h, m, l = 0;
                const float x0 = e < 2 ? b0[2 * e] : b1[2 * e - 4], x1 = e < 2 ? b0[2 * e + 1] : b1[2 * e - 3];
                if (PAIR)
                    split2h(x0 * kPairScale, x1 * kPairScale, h, m);
                else
                    split2(x0, x1, h, m, l);
                bh[e] = h, bm[e] = m, bl[e] = l;
            }
            if (PAIR) {   // planes (h, m) = (hi, lo) of fp16: leading product + the two corrections
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    lo[j] = FP16::mfma32(__builtin_bit_cast(f16x8_t, am[j]), __builtin_bit_cast(f16x8_t, bh), lo[j]);
                    lo[j] = FP16::mfma32(__builtin_bit_cast(f16x8_t, ah[j]), __builtin_bit_cast(f16x8_t, bm), lo[j]);
                    acc[j] = FP16::mfma32(__builtin_bit_cast(f16x8_t, ah[j]), __builtin_bit_cast(f16x8_t, bh), acc[j]);
                }
                continue;
            }
#define DIR_MM(ACC, A, B)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) ACC[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(         \
        __builtin_bit_cast(bf16x8_t, A[j]), __builtin_bit_cast(bf16x8_t, B), ACC[j], 0, 0, 0)
            DIR_MM(lo, al, bh);     // same order of products and accumulators as sim_split_kernel
            DIR_MM(lo, ah, bl);
            DIR_MM(lo, am, bm);
            DIR_MM(lo, am, bh);
            DIR_MM(lo, ah, bm);
            DIR_MM(acc, ah, bh);
#undef DIR_MM
        }
        cur = cur + 1 == kStages ? 0 : cur + 1;
    }

    const int n = i0 + wave * 32 + lrow;
    if (n < NP) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = qb * kQB + j * 32 + 8 * g + 4 * lhi + e;
                    const float v = acc[j][4 * g + e] + lo[j][4 * g + e];
                    if (q < NQ) out[(size_t)q * ldo + n] = PAIR ? v * (1.f / (kPairScale * kPairScale)) : v;
                }
    }
}

// ---- PCA whitening of a LARGE descriptor set on the same machinery -------------------------------------------------------
//   out[n][j] = alpha[j] * sum_k (X[n][k] - mean[k]) * C[j][k]        (dirtorch/utils/common.py:221-232: pca.transform of the
//                                                                      database before scoring, test_dir.py:136-138)
// At config D's size (N = 1 006 322, K = v = 2048) this is 8.4 TFLOP: 85 ms on the exact fp32 MFMA chain of gemm_f32.hip
// (157 TFLOP/s peak), 40x the similarity + ranking step it feeds.  Both operands are bounded - L2-normalised descriptors minus
// their mean, orthonormal PCA rows - so the PAIR form above applies: two fp16 planes of 2^10 x per operand, three products,
// fp32 accumulation.  What differs from sim_split_lc_kernel<true>:
//   * X takes the database's place (streamed once per component block by LDS-DMA as raw fp32, 256 rows per workgroup) and the
//     mean is subtracted in fp32 right before the split, as the reference subtracts it before its GEMM (no cancellation between
//     two large dot products); the mean vector lives in LDS (K floats);
//   * the components take the queries' place (their plane image: split_queries_kernel<true>), 96 rows per block - but there are
//     v / 96 = 22 blocks, not one, so the grid is 1-D with an XCD-aware decode: the 22 workgroups that share an X tile run on the
//     SAME XCD at the same time and walk K in the same rotated order - one of them misses to HBM, 21 hit that XCD's L2;
//   * the result is stored TRANSPOSED, out[n][j] row-major (16-byte stores: a lane holds four consecutive j of one row n),
//     scaled by 2^-20 alpha[j].
template <bool HAS_MEAN>
__global__ void __launch_bounds__(768) whiten_split_kernel(const float* __restrict__ P, int ldp, int NP, int K,
                                                          const uint16_t* __restrict__ img, const float* __restrict__ mean,
                                                          const float* __restrict__ alpha, float* __restrict__ out, int ldo,
                                                          int NQ, int T, int tiles, int qblocks) {
    constexpr int kSlabQ = dir::kSlabQ2;
    constexpr int kStage = dir::kStage2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lmean = (float*)(smem + kStages * kStage);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // hardware places block id on XCD id % 8: slot (id / 8) of an XCD = (tile group, component block), blocks fastest
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = (slot / qblocks) * 8 + xcd, qb = slot % qblocks;
    if (tile >= tiles) return;
    const int i0 = tile * kRowsP;
    const int rows = min(kRowsP, NP - i0);
    const int rot = (int)(((unsigned)tile * 7u) % (unsigned)T);
    if (HAS_MEAN)   // the table holds 2^10 mean: (x - mean) 2^10 is then ONE fma per value, bit-identical (a power-of-two scale commutes with rounding)
        for (int i = tid; i < K; i += 768) lmean[i] = mean[i] * kPairScale;

    if (wave >= 8) {
        // ================================ loaders (as sim_split_lc_kernel<true>) ================================
        const int lw = wave - 8;
        const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(P + (size_t)i0 * ldp), 0, (int)((((size_t)rows - 1) * ldp + K) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_q = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const char*)img + (size_t)qb * T * kSlabQ), 0, T * kSlabQ, 0x00020000);
        uint32_t pvoff[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (lw * 8 + i) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            pvoff[i] = (uint32_t)row * (uint32_t)ldp * 4u + (uint32_t)chunk * 16u;   // rows past `rows`: out of range -> 0
        }
        const int q0 = lw * 3;
        auto issue = [&](int t) __attribute__((always_inline)) {
            int u = t + rot;
            u = u >= T ? u - T : u;
            char* stage = smem + (t % kStages) * kStage;
#pragma unroll
            for (int i = 0; i < 8; ++i) dma16s(rsrc_p, stage + (lw * 8 + i) * 1024, pvoff[i], (uint32_t)u * 128u);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                dma16s(rsrc_q, stage + kSlabP + (q0 + i) * 1024, (uint32_t)((q0 + i) * 1024 + lane * 16), (uint32_t)u * kSlabQ);
        };
        issue(0);
        if (T > 1) issue(1);
        for (int t = 0; t < T; ++t) {
            if (t + 1 < T)
                asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            ring_barrier();   // hand-off t: slab t is complete (and, at t = 0, the mean table); the consumers have left slab t - 1
            if (t + 2 < T) issue(t + 2);
        }
        return;
    }

    // ==================================== consumers =============================================================
    const int lrow = lane & 31, lhi = lane >> 5;
    f32x16_t acc[3], lo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f, lo[j][e] = 0.f;
    const int boff = (wave * 32 + lrow) * 128, bswz = (lrow >> 1) & 7;
    const int aoff = kSlabP + lrow * 64, aswz = (lrow >> 2) & 3;

    int cur = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's share of the mean table is written)
    for (int t = 0; t < T; ++t) {
        ring_barrier();   // hand-off t (see the loaders)
        const char* stage = smem + cur * kStage;
        int u = t + rot;
        u = u >= T ? u - T : u;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c0 = s * 4 + lhi * 2;
            f32x4_t b0 = *(const f32x4_t*)(stage + boff + ((c0 ^ bswz) << 4));
            f32x4_t b1 = *(const f32x4_t*)(stage + boff + (((c0 + 1) ^ bswz) << 4));
            if (HAS_MEAN) {   // k = 32 u + 16 s + 8 lhi + 0..7
                const f32x4_t m0 = *(const f32x4_t*)(lmean + u * 32 + c0 * 4), m1 = *(const f32x4_t*)(lmean + u * 32 + c0 * 4 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) b0[e] = __builtin_fmaf(b0[e], kPairScale, -m0[e]), b1[e] = __builtin_fmaf(b1[e], kPairScale, -m1[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) b0[e] *= kPairScale, b1[e] *= kPairScale;
            }
            u32x4_t ah[3], al[3];
            const int ach = ((s * 2 + lhi) ^ aswz) << 4;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                ah[j] = *(const u32x4_t*)(stage + aoff + 0 * kPlane + j * 2048 + ach);
                al[j] = *(const u32x4_t*)(stage + aoff + 1 * kPlane + j * 2048 + ach);
            }
            u32x4_t bh, bl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t h, l;
                const float x0 = e < 2 ? b0[2 * e] : b1[2 * e - 4], x1 = e < 2 ? b0[2 * e + 1] : b1[2 * e - 3];
                split2h(x0, x1, h, l);      // (already scaled by 2^10)
                bh[e] = h, bl[e] = l;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                lo[j] = FP16::mfma32(__builtin_bit_cast(f16x8_t, al[j]), __builtin_bit_cast(f16x8_t, bh), lo[j]);
                lo[j] = FP16::mfma32(__builtin_bit_cast(f16x8_t, ah[j]), __builtin_bit_cast(f16x8_t, bl), lo[j]);
                acc[j] = FP16::mfma32(__builtin_bit_cast(f16x8_t, ah[j]), __builtin_bit_cast(f16x8_t, bh), acc[j]);
            }
        }
        cur = cur + 1 == kStages ? 0 : cur + 1;
    }

    // D[i][j]: lane holds X row lrow of the strip and component rows i = 32 j + 8 g + 4 lhi + e: four consecutive outputs of row n
    const int n = i0 + wave * 32 + lrow;
    if (n < NP) {
        float* orow = out + (size_t)n * ldo;
        const bool vec = (ldo & 3) == 0 && (((uintptr_t)out) & 15) == 0;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int q = qb * kQB + j * 32 + 8 * g + 4 * lhi;
                if (q >= NQ) continue;
                f32x4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float al_ = (alpha && q + e < NQ) ? alpha[q + e] : 1.f;
                    v[e] = (acc[j][4 * g + e] + lo[j][4 * g + e]) * (al_ * (1.f / (kPairScale * kPairScale)));
                }
                if (vec && q + 3 < NQ) {
                    *(f32x4_t*)(orow + q) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (q + e < NQ) orow[q + e] = v[e];
                }
            }
    }
}

size_t whiten_split_workspace_bytes(int v, int K) { return (size_t)ceil_div(v, kQB) * (size_t)ceil_div(K, 32) * kSlabQ2; }

bool whiten_split_admissible(const float* X, int ldx, int N, int K, int v) {
    return N > 0 && v > 0 && K > 0 && (K % 32) == 0 && (ldx % 4) == 0 && ((uintptr_t)X & 15) == 0 && ldx >= K &&
           (size_t)ldx * 4 * kRowsP < (1ull << 31) && (size_t)ceil_div(K, 32) * kSlabQ2 < (1ull << 31) &&
           kLds2 + (size_t)K * 4 <= 160 * 1024 && (size_t)ceil_div(N, kRowsP) * ceil_div(v, kQB) < (1ull << 30);
}

// X [N, K] (row stride ldx), components [v, K] (row stride ldc), mean [K] or null, alpha [v] or null -> out [N, v] (row stride ldo).
// Operands must lie in (-64, 64) (X - mean and the components): the caller's promise, as for similarity_split(unit_range = true).
int whiten_split(const float* X, int ldx, int N, const float* comps, int ldc, int v, int K, const float* mean, const float* alpha,
                 float* out, int ldo, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!whiten_split_admissible(X, ldx, N, K, v))
        return fail(DIR_ERR_INVALID, "whiten_split: needs K % 32 == 0, K <= 6144, ldx % 4 == 0 and a 16-byte aligned X");
    if (ldc < K || ldo < v) return fail(DIR_ERR_INVALID, "whiten_split: ldc >= K and ldo >= v");
    if (!workspace || workspace_bytes < whiten_split_workspace_bytes(v, K))
        return fail(DIR_ERR_WORKSPACE, "whiten_split: workspace too small");
    if (((uintptr_t)workspace & 15) != 0) return fail(DIR_ERR_INVALID, "whiten_split: workspace must be 16-byte aligned");
    const int T = K / 32, qblocks = ceil_div(v, kQB), tiles = ceil_div(N, kRowsP);
    const int lds = kLds2 + K * 4;
    static std::atomic<uint64_t> attr_m{0}, attr_n{0};
    DIR_HIP_CHECK(ensure_dynamic_lds((const void*)whiten_split_kernel<true>, 160 * 1024, attr_m));
    DIR_HIP_CHECK(ensure_dynamic_lds((const void*)whiten_split_kernel<false>, 160 * 1024, attr_n));
    hipLaunchKernelGGL(split_queries_kernel<true>, dim3(T, qblocks), dim3(256), 0, stream, comps, ldc, v, K, (uint16_t*)workspace);
    const unsigned grid = (unsigned)(8 * ceil_div(tiles, 8) * qblocks);
    if (mean)
        hipLaunchKernelGGL(whiten_split_kernel<true>, dim3(grid), dim3(768), lds, stream, X, ldx, N, K, (const uint16_t*)workspace, mean,
                           alpha, out, ldo, v, T, tiles, qblocks);
    else
        hipLaunchKernelGGL(whiten_split_kernel<false>, dim3(grid), dim3(768), lds, stream, X, ldx, N, K, (const uint16_t*)workspace, mean,
                           alpha, out, ldo, v, T, tiles, qblocks);
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

size_t similarity_split_workspace_bytes(int NQ, int K) {
    return (size_t)ceil_div(NQ, kQB) * (size_t)ceil_div(K, 32) * kSlabQ;
}

bool similarity_split_admissible(const float* P, int ldp, const float* Q, int ldq, int NP, int NQ, int K) {
    (void)Q, (void)ldq;
    return NP > 0 && NQ > 0 && K > 0 && (K % 32) == 0 && (ldp % 4) == 0 && ((uintptr_t)P & 15) == 0 &&
           (size_t)ldp * 4 * kRowsP < (1ull << 31) && (size_t)ceil_div(K, 32) * kSlabQ < (1ull << 31) &&
           ceil_div(NQ, kQB) < 65536;
}

int similarity_split(const float* P, int ldp, const float* Q, int ldq, float* out, int ldo, int NP, int NQ, int K,
                     void* workspace, size_t workspace_bytes, hipStream_t stream, bool unit_range) {
    if (!similarity_split_admissible(P, ldp, Q, ldq, NP, NQ, K))
        return fail(DIR_ERR_INVALID, "similarity_split: needs K % 32 == 0, ldp % 4 == 0 and a 16-byte aligned database");
    if (ldp < K || ldq < K || ldo < NP) return fail(DIR_ERR_INVALID, "similarity_split: ldp, ldq >= K and ldo >= NP");
    if (!workspace || workspace_bytes < similarity_split_workspace_bytes(NQ, K))
        return fail(DIR_ERR_WORKSPACE, "similarity_split: workspace too small");
    if (((uintptr_t)workspace & 15) != 0) return fail(DIR_ERR_INVALID, "similarity_split: workspace must be 16-byte aligned");
    static std::atomic<uint64_t> attr_done{0}, attr_done_lc{0}, attr_done_pair{0};
    DIR_HIP_CHECK(ensure_dynamic_lds((const void*)sim_split_kernel, kLds, attr_done));
    DIR_HIP_CHECK(ensure_dynamic_lds((const void*)sim_split_lc_kernel<false>, kLds, attr_done_lc));
    DIR_HIP_CHECK(ensure_dynamic_lds((const void*)sim_split_lc_kernel<true>, kLds2, attr_done_pair));
    const bool v1 = env().sim_v1;   // A/B and bisecting: the one-role kernel
    const int T = K / 32, qblocks = ceil_div(NQ, kQB);
    if (unit_range && !v1) {   // operands in (-64, 64): two fp16 planes, three products
        hipLaunchKernelGGL(split_queries_kernel<true>, dim3(T, qblocks), dim3(256), 0, stream, Q, ldq, NQ, K, (uint16_t*)workspace);
        hipLaunchKernelGGL(sim_split_lc_kernel<true>, dim3(ceil_div(NP, kRowsP), qblocks), dim3(768), kLds2, stream, P, ldp, NP, K,
                           (const uint16_t*)workspace, out, ldo, NQ, T);
        DIR_HIP_CHECK(hipGetLastError());
        return DIR_OK;
    }
    hipLaunchKernelGGL(split_queries_kernel<false>, dim3(T, qblocks), dim3(256), 0, stream, Q, ldq, NQ, K, (uint16_t*)workspace);
    if (v1)
        hipLaunchKernelGGL(sim_split_kernel, dim3(ceil_div(NP, kRowsP), qblocks), dim3(512), kLds, stream, P, ldp, NP, K,
                           (const uint16_t*)workspace, out, ldo, NQ, T);
    else
        hipLaunchKernelGGL(sim_split_lc_kernel<false>, dim3(ceil_div(NP, kRowsP), qblocks), dim3(768), kLds, stream, P, ldp, NP, K,
                           (const uint16_t*)workspace, out, ldo, NQ, T);
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

}  // namespace dir
