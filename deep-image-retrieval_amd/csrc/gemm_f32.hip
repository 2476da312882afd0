// gemm_f32.hip — exact-fp32 "NT" GEMM on the matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
//   out[j][i] = alpha[i] * sum_k P[i][k] * (Q[j][k] - qsub[k]) + bias[i]
//
// One kernel serves the three fp32 contractions of the descriptor path:
//   FC "whitening"   P = fc.weight [D_out, 2048], Q = pooled features [B, 2048], bias = fc.bias
//                    (dirtorch/nets/rmac_resnet.py:34,66)
//   PCA whitening    P = components_[:v], Q = descriptors, qsub = mean_, alpha = 1/(m * var^p)
//                    (dirtorch/utils/common.py:221-232)
//   similarity       P = database descriptors [N, D], Q = queries [Q, D]  -> scores [Q, N]
//                    (dirtorch/utils/common.py:30-38)
// The f32 MFMA is bit-for-bit a k-ordered fmaf chain, so results are fp32-exact up to summation
// order.  P is the long operand: 128 P rows per workgroup, one 32-row strip per wave; all of a
// Q tile (32*TJ rows) is shared by the four waves.  The similarity case is HBM-bound on P
// (N*D*4 bytes read once); the tile loop keeps two K-slabs of 32 floats in LDS (register staged
// so that `- qsub[k]` and the K tail are folded into the staging pass).
#include "dir_common.h"
#include "pointwise.h"

namespace dir {

template <int TJ>
__global__ void __launch_bounds__(256) gemm_nt_f32_kernel(const float* __restrict__ P, int ldp,
                                                         const float* __restrict__ Q, int ldq,
                                                         float* __restrict__ out, int ldo, int NP,
                                                         int NQ, int K,
                                                         const float* __restrict__ qsub,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ alpha,
                                                         int tiles_i, int vec_ok, int ksplit,
                                                         float* __restrict__ partial) {
    constexpr int BI = 128, BJ = 32 * TJ;
    constexpr int PS = BI * 128;              // bytes of one P slab (128 rows x 32 floats)
    constexpr int STAGE_BYTES = (BI + BJ) * 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_i = wg % tiles_i;  // i fastest: consecutive blocks of an XCD share the Q tile
    const int tile_j = wg / tiles_i;
    const int i0 = tile_i * BI, j0 = tile_j * BJ;

    const int slot = tid & 7;
    const int srcchunk = slot ^ ((tid >> 4) & 7);  // 16-byte chunk (4 floats) of the 128-byte row
    const int prow = tid >> 3;                     // + 32 * i

    f32x4_t pr[4], qr[TJ];
    // vec_ok (K, ldp, ldq multiples of 4 and 16-byte aligned bases): 16-byte loads.  Otherwise - an odd
    // descriptor width such as --whitenv 50 (dirtorch/test_dir.py:210, common.py:226) - the same slab
    // is gathered element by element with the K tail zero-filled; the arithmetic is unchanged.
    auto load4 = [&](const float* base, size_t row_off, int k, bool ok) -> f32x4_t {
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
        if (!ok) return v;
        if (vec_ok) return *(const DIR_GLOBAL f32x4_t*)(base + row_off + k);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k + e < K) v[e] = base[row_off + k + e];
        return v;
    };
    auto fetch = [&](int k0) {
        const int k = k0 + srcchunk * 4;
        const bool kok = k < K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i0 + i * 32 + prow;
            pr[i] = load4(P, (size_t)row * ldp, k, kok && row < NP);
        }
        const f32x4_t sub = load4(qsub, 0, k, qsub != nullptr && kok);
#pragma unroll
        for (int i = 0; i < TJ; ++i) {
            const int row = j0 + i * 32 + prow;
            const bool ok = kok && row < NQ;
            f32x4_t v = load4(Q, (size_t)row * ldq, k, ok);
            if (ok) v = v - sub;
            if (!vec_ok && ok) {   // keep the zero-filled K tail zero (0 - sub would leak the mean in)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k + e >= K) v[e] = 0.f;
            }
            qr[i] = v;
        }
    };
    auto commit = [&](char* stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4_t*)(stage + (i * 256 + tid) * 16) = pr[i];
#pragma unroll
        for (int i = 0; i < TJ; ++i) *(f32x4_t*)(stage + PS + (i * 256 + tid) * 16) = qr[i];
    };

    const int lrow = lane & 31, lhi = lane >> 5, lswz = (lane >> 1) & 7;
    int loff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) loff[c] = lrow * 128 + (((2 * c + lhi) ^ lswz) << 4);
    const int pbase = (wave * 32) * 128;

    f32x16_t acc[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    auto compute = [&](const char* stage) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4_t pf = *(const f32x4_t*)(stage + pbase + loff[c]);
            f32x4_t qf[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) qf[j] = *(const f32x4_t*)(stage + PS + j * 4096 + loff[c]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[q], qf[j][q], acc[j], 0, 0, 0);
        }
    };

    // split-K (blockIdx.y = slice): few output tiles and a long K - the FC of a small batch, 16 tiles x 2048 - would
    // leave most CUs idle; each slice writes its raw sums to partial[slice][NQ][NP], gemm_splitk_finalize_kernel adds
    // them in slice order with alpha / bias
    const int Tall = (K + 31) / 32;
    const int Ts = (Tall + ksplit - 1) / ksplit;
    const int t0 = (int)blockIdx.y * Ts;
    const int T = min(Ts, Tall - t0);
    // Every workgroup walks K from a different starting slab (a rotation of the same sum).  With a
    // power-of-two row pitch (D = 2048 floats = 8 KiB) workgroups marching through K in lock-step
    // would all be fetching addresses congruent modulo the pitch - the same few HBM channels - at
    // any instant; staggering the phase spreads the stream over all of them.
    const int rot = T > 0 ? (int)(((unsigned)tile_i * 7u + (unsigned)tile_j * 3u) % (unsigned)T) : 0;
    auto slab = [&](int t) {
        int u = t + rot;
        if (u >= T) u -= T;
        return (t0 + u) * 32;
    };
    char* stage0 = smem;
    char* stage1 = smem + STAGE_BYTES;
    if (T > 0) {
        fetch(slab(0));
        commit(stage0);
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        char* cur = (t & 1) ? stage1 : stage0;
        char* nxt = (t & 1) ? stage0 : stage1;
        const bool more = t + 1 < T;
        if (more) fetch(slab(t + 1));
        compute(cur);
        if (more) commit(nxt);
        __syncthreads();
    }

    // D[i][j]: lane holds column j = lane & 31, rows i = 8*g + 4*(lane >> 5) + 0..3 for g = 0..3
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int jj = j0 + j * 32 + lrow;
        if (jj >= NQ) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ii = i0 + wave * 32 + 8 * g + 4 * lhi;
            float v[4] = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ii + e < NP && ksplit == 1) {
                    float r = v[e];
                    if (alpha) r *= alpha[ii + e];
                    if (bias) r += bias[ii + e];
                    v[e] = r;
                }
            }
            float* dst = ksplit == 1 ? out + (size_t)jj * ldo + ii
                                     : partial + ((size_t)blockIdx.y * NQ + jj) * (size_t)NP + ii;
            if (ii + 3 < NP && ksplit == 1 && ((ldo & 3) == 0) && (((uintptr_t)out & 15) == 0)) {
                *(DIR_GLOBAL f32x4_t*)dst = (f32x4_t){v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ii + e < NP) dst[e] = v[e];
            }
        }
    }
}

// Few Q rows (batch-sized FC: NQ <= 32): the MFMA tiling would leave 240 of 256 CUs idle, and the
// job is a pure stream of P (NP*K*4 bytes, read once).  One wave per P row: the row lives in
// registers (K/64 floats per lane, coalesced float4 loads), the NQ small Q rows come from L2, each
// dot product is an fmaf chain + a 64-lane butterfly.  HBM-bound on P.
template <int KV>  // float4 per lane: K <= KV * 256
__global__ void __launch_bounds__(256) gemm_nt_small_kernel(const float* __restrict__ P, int ldp,
                                                           const float* __restrict__ Q, int ldq,
                                                           float* __restrict__ out, int ldo, int NP,
                                                           int NQ, int K,
                                                           const float* __restrict__ qsub,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ alpha) {
    // two P rows per wave: every Q row fetched from L2 feeds two dot products
    const int lane = threadIdx.x & 63;
    const int i0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (i0 >= NP) return;
    const bool two = i0 + 1 < NP;
    f32x4_t pa[KV], pb[KV], sv[KV];
#pragma unroll
    for (int c = 0; c < KV; ++c) {
        const int k = (c * 64 + lane) * 4;
        pa[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        pb[c] = pa[c];
        sv[c] = pa[c];
        if (k < K) {
            pa[c] = *(const DIR_GLOBAL f32x4_t*)(P + (size_t)i0 * ldp + k);
            if (two) pb[c] = *(const DIR_GLOBAL f32x4_t*)(P + (size_t)(i0 + 1) * ldp + k);
            if (qsub) sv[c] = *(const DIR_GLOBAL f32x4_t*)(qsub + k);
        }
    }
    const float al0 = alpha ? alpha[i0] : 1.f, al1 = (alpha && two) ? alpha[i0 + 1] : 1.f;
    const float bi0 = bias ? bias[i0] : 0.f, bi1 = (bias && two) ? bias[i0 + 1] : 0.f;
    for (int j = 0; j < NQ; ++j) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int c = 0; c < KV; ++c) {
            const int k = (c * 64 + lane) * 4;
            if (k < K) {
                const f32x4_t q = *(const DIR_GLOBAL f32x4_t*)(Q + (size_t)j * ldq + k) - sv[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s0 = fmaf(pa[c][e], q[e], s0);
                    s1 = fmaf(pb[c][e], q[e], s1);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s0 += __shfl_xor(s0, off, 64);
            s1 += __shfl_xor(s1, off, 64);
        }
        if (lane == 0) {
            out[(size_t)j * ldo + i0] = al0 * s0 + bi0;
            if (two) out[(size_t)j * ldo + i0 + 1] = al1 * s1 + bi1;
        }
    }
}

__global__ void __launch_bounds__(256) gemm_splitk_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                  int ldo, int NP, int NQ, int ksplit,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ alpha) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)NP * NQ) return;
    const int j = (int)(idx / NP), i = (int)(idx - (long)j * NP);
    float r = 0.f;
    for (int z = 0; z < ksplit; ++z) r += partial[((size_t)z * NQ + j) * (size_t)NP + i];   // fixed order: run-to-run identical
    if (alpha) r *= alpha[i];
    if (bias) r += bias[i];
    out[(size_t)j * ldo + i] = r;
}

// How many K slices the MFMA path runs for a shape (1 = no split): only when the output has too few 128 x 32TJ tiles
// to occupy the chip and K is long enough that every slice keeps >= 4 slabs of 32.
int gemm_splitk_factor(int NP, int NQ, int K) {
    const int tj = NQ >= 97 ? 4 : (NQ >= 65 ? 3 : (NQ >= 33 ? 2 : 1));
    const long tiles = (long)ceil_div(NP, 128) * ceil_div(NQ, 32 * tj);
    const int T = ceil_div(K, 32);
    if (tiles >= 128 || T < 16 || NQ <= 4) return 1;   // a handful of Q rows: the one-wave-per-two-rows kernel below
    long s = 256 / tiles;
    if (s > 16) s = 16;
    if (s > T / 4) s = T / 4;
    while (s > 1 && (size_t)s * NQ * NP * sizeof(float) > kGemmSplitKMaxBytes) --s;
    return s < 2 ? 1 : (int)s;
}

int gemm_nt_f32(const float* P, int ldp, const float* Q, int ldq, float* out, int ldo, int NP,
                int NQ, int K, const float* qsub, const float* bias, const float* alpha,
                hipStream_t stream, float* scratch, size_t scratch_bytes) {
    if (NP <= 0 || NQ <= 0) return DIR_OK;
    // the one-wave-per-two-rows kernel re-reads every Q row from L2 for every wave: fine for a handful of Q rows,
    // 268 MB of L2 traffic for the FC of a 32-image batch (87 us) - those go to the MFMA path with split-K
    const bool few_q = gemm_splitk_factor(NP, NQ, K) == 1;
    if (few_q && NQ <= 32 && K <= 2048 && K > 0 && !(K & 3) && !(ldp & 3) && !(ldq & 3) &&
        !((uintptr_t)P & 15) && !((uintptr_t)Q & 15) && !(qsub && ((uintptr_t)qsub & 15))) {
        const unsigned blocks = (unsigned)ceil_div(NP, 8);   // 4 waves x 2 rows
        if (K <= 1024)
            hipLaunchKernelGGL(gemm_nt_small_kernel<4>, dim3(blocks), dim3(256), 0, stream, P, ldp,
                               Q, ldq, out, ldo, NP, NQ, K, qsub, bias, alpha);
        else
            hipLaunchKernelGGL(gemm_nt_small_kernel<8>, dim3(blocks), dim3(256), 0, stream, P, ldp,
                               Q, ldq, out, ldo, NP, NQ, K, qsub, bias, alpha);
        DIR_HIP_CHECK(hipGetLastError());
        return DIR_OK;
    }
    if (K <= 0 || ldp < K || ldq < K || ldo < NP)
        return fail(DIR_ERR_INVALID, "gemm_nt_f32: K must be positive and ldp, ldq >= K, ldo >= NP");
    if (((uintptr_t)P & 3) || ((uintptr_t)Q & 3) || ((uintptr_t)out & 3) || (qsub && ((uintptr_t)qsub & 3)))
        return fail(DIR_ERR_INVALID, "gemm_nt_f32: operands must be 4-byte aligned");
    // 16-byte loads when the layout allows them; any other K / pitch takes the element-wise gather
    const int vec_ok = !(K & 3) && !(ldp & 3) && !(ldq & 3) && !((uintptr_t)P & 15) && !((uintptr_t)Q & 15) &&
                       !(qsub && ((uintptr_t)qsub & 15));
    const int tiles_i = ceil_div(NP, 128);
    // Q tile: as wide as needed up to 128 rows, then loop tiles over j.
    int tj = NQ >= 97 ? 4 : (NQ >= 65 ? 3 : (NQ >= 33 ? 2 : 1));
    const int tiles_j = ceil_div(NQ, 32 * tj);
    const long nblk = (long)tiles_i * tiles_j;
    if (nblk >= (1L << 31)) return fail(DIR_ERR_INVALID, "gemm_nt_f32: grid too large");
    int ksplit = gemm_splitk_factor(NP, NQ, K);
    float* partial = nullptr;
    bool own = false;
    if (ksplit > 1) {
        const size_t need = (size_t)ksplit * NQ * NP * sizeof(float);
        if (scratch && scratch_bytes >= need) {
            partial = scratch;
        } else if (hipMallocAsync((void**)&partial, need, stream) == hipSuccess) {
            own = true;   // callers without a workspace (the C-ABI entry points): stream-ordered scratch
        } else {          // no memory pool on this device / out of memory: the unsplit form is always available
            (void)hipGetLastError();
            partial = nullptr;
            ksplit = 1;
        }
    }
#define DIR_G(TJ)                                                                                        \
    hipLaunchKernelGGL(gemm_nt_f32_kernel<TJ>, dim3((unsigned)nblk, (unsigned)ksplit), dim3(256), 0, stream, P, \
                       ldp, Q, ldq, out, ldo, NP, NQ, K, qsub, bias, alpha, tiles_i, vec_ok, ksplit, partial)
    switch (tj) {
        case 1: DIR_G(1); break;
        case 2: DIR_G(2); break;
        case 3: DIR_G(3); break;
        default: DIR_G(4); break;
    }
#undef DIR_G
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && ksplit > 1) {
        const long total = (long)NP * NQ;
        hipLaunchKernelGGL(gemm_splitk_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                           partial, out, ldo, NP, NQ, ksplit, bias, alpha);
        e = hipGetLastError();
    }
    if (own) {
        const hipError_t fe = hipFreeAsync(partial, stream);
        if (e == hipSuccess) e = fe;
    }
    DIR_HIP_CHECK(e);
    return DIR_OK;
}

}  // namespace dir
