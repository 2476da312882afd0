// ranking.hip — device-side ranking for the revisited Oxford/Paris AP protocol.
//
// The reference downloads the Q x N score matrix and runs np.argsort over every row three times
// (dirtorch/datasets/generic.py:196-224); at N = 10^6 that is 280 MB of D2H and 210 million-element
// sorts.  AP only needs the rank of each positive among the non-junk images, i.e.
//     rank(p) = #{ j : j ranks before p } - #{ junk j : j ranks before p },
// where "j ranks before p" is (s_j > s_p) or (s_j == s_p and j > p) - the order of
// np.argsort(scores)[::-1] with ties resolved by descending index.  The first term is a dense count
// over the row (rank_counts below, HBM-bound: each score read once per 4096 probes); the second involves
// only the few hundred listed images of the query (revisitop_ap_kernel at the end of this file).
#include "dir_common.h"
#include "pointwise.h"

#include <algorithm>

namespace dir {

// Work: the count of a probe depends on that probe and the row only, and "j ranks before p" is a comparison of the
// 64-bit keys (score mapped to an order-preserving uint32, index): key_j > key_p.  With the P probes of a query SORTED
// by key, an item ranks before exactly the first k_j = lower_bound(key_j) of them, so one binary search per score
// (log2 P LDS reads) and a histogram of k_j replace the P comparisons per score of a direct count:
//     count[probe at sorted position i] = #{ j : k_j > i } = sum_{k > i} hist[k].
//   rank_sort_kernel      one workgroup per (query, slice of <= 4096 probes): bitonic sort of (key, slot) in LDS;
//                         writes the permutation into the probe_scores slice (as integers - no scratch buffer)
//   rank_hist_kernel      one workgroup per (16384 scores, query): keys of the slice rebuilt through the permutation
//                         into LDS, every score read ONCE (coalesced), binary search, LDS histogram, non-zero bins
//                         added to counts[q][sorted position] (bin k lives at k - 1; bin 0 is never needed)
//   rank_finalize_kernel  one workgroup per (query, slice): suffix sums of the bins, scattered to the probes' own
//                         slots; probe_scores gets scores[q][probe]
// HBM-bound: N*4 B per query per slice of probes.  A NaN score ranks before nothing and nothing ranks before a NaN
// probe (every comparison of the reference's ordering is false); -0 == +0 as in the float compare.
constexpr int kRankChunk = 16384;  // scores per workgroup
constexpr int kMaxProbes = 4096;   // probes per query per launch (keys 32 KiB + bins 16 KiB of LDS)
constexpr uint64_t kKeyMax = ~0ull;

__device__ __forceinline__ uint64_t rank_key(float s, int idx) {
    if (s == 0.f) s = 0.f;                               // -0 -> +0: they compare equal
    const uint32_t b = __builtin_bit_cast(uint32_t, s);
    const uint32_t u = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // order-preserving for every non-NaN float
    return ((uint64_t)u << 32) | (uint32_t)idx;
}

// probe_idx / perm_out point at the slice: [Q][ldp] rows, P (<= kMaxProbes) columns in use, P2 = pow2 >= P.
__global__ void __launch_bounds__(256) rank_sort_kernel(const float* __restrict__ scores, int lds,
                                                       const int* __restrict__ probe_idx, int P, int P2, int ldp,
                                                       int* __restrict__ perm_out) {
    extern __shared__ __attribute__((aligned(16))) char rsm[];
    uint64_t* keys = (uint64_t*)rsm;
    int* slot = (int*)(rsm + (size_t)P2 * 8);
    const int q = blockIdx.x;
    const float* row = scores + (size_t)q * lds;
    for (int i = threadIdx.x; i < P2; i += 256) {
        const int idx = i < P ? probe_idx[(size_t)q * ldp + i] : -1;
        uint64_t k = kKeyMax;                            // unused slots and NaN probes sort last: their suffix sums are 0
        if (idx >= 0) {
            const float s = row[idx];
            if (s == s) k = rank_key(s, idx);
        }
        keys[i] = k;
        slot[i] = i;
    }
    __syncthreads();
    for (int len = 2; len <= P2; len <<= 1)
        for (int st = len >> 1; st > 0; st >>= 1) {
            for (int i = threadIdx.x; i < P2 / 2; i += 256) {
                const int lo = ((i / st) * st * 2) + (i % st), hi = lo + st;
                const bool up = ((lo & len) == 0);
                const uint64_t a = keys[lo], b = keys[hi];
                const int sa = slot[lo], sb = slot[hi];
                // ties between equal keys (a probe listed twice, the padding) break on the slot: a total order
                const bool gt = a > b || (a == b && sa > sb);
                if (gt == up) {
                    keys[lo] = b, keys[hi] = a;
                    slot[lo] = sb, slot[hi] = sa;
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < P; i += 256) perm_out[(size_t)q * ldp + i] = slot[i];   // padding sorts behind every real slot
}

__global__ void __launch_bounds__(256) rank_hist_kernel(const float* __restrict__ scores, int lds, int N,
                                                       const int* __restrict__ probe_idx, int P, int P2, int ldp,
                                                       const int* __restrict__ perm, int* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) char rsm[];
    uint64_t* keys = (uint64_t*)rsm;
    int* hist = (int*)(rsm + (size_t)P2 * 8);
    const int q = blockIdx.y;
    const float* row = scores + (size_t)q * lds;
    for (int i = threadIdx.x; i < P2; i += 256) {
        uint64_t k = kKeyMax;
        if (i < P) {
            const int sl = perm[(size_t)q * ldp + i];
            const int idx = probe_idx[(size_t)q * ldp + sl];
            if (idx >= 0) {
                const float s = row[idx];
                if (s == s) k = rank_key(s, idx);
            }
        }
        keys[i] = k;
        hist[i] = 0;
    }
    __syncthreads();
    const uint64_t kmin = keys[0];
    const long j0 = (long)blockIdx.x * kRankChunk;
    constexpr int U = 8;
    for (int base = 0; base < kRankChunk; base += 256 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = j0 + base + u * 256 + threadIdx.x;
            v[u] = j < N ? row[j] : NAN;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = j0 + base + u * 256 + threadIdx.x;
            if (!(v[u] == v[u])) continue;               // NaN (and the tail past N) ranks before nothing
            const uint64_t kj = rank_key(v[u], (int)j);
            if (kj <= kmin) continue;                    // before no probe: bin 0
            int pos = 0;                                 // lower bound in a power-of-two table
            for (int st = P2 >> 1; st > 0; st >>= 1)
                if (keys[pos + st - 1] < kj) pos += st;
            // pos >= 1 here.  The last table entry is never passed unless it is a real key below kj
            if (pos < P2 && keys[pos] < kj) ++pos;
            atomicAdd(&hist[pos - 1], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += 256)
        if (hist[i]) atomicAdd(counts + (size_t)q * ldp + i, hist[i]);
}

__global__ void __launch_bounds__(256) rank_finalize_kernel(const float* __restrict__ scores, int lds,
                                                           const int* __restrict__ probe_idx, int P, int ldp,
                                                           int* __restrict__ counts, float* __restrict__ probe_scores) {
    extern __shared__ __attribute__((aligned(16))) char rsm[];
    int* bins = (int*)rsm;             // [P] bin k at k - 1; becomes the suffix sums
    int* slot = bins + P;              // [P] sorted position -> slot
    __shared__ int carry;
    const int q = blockIdx.x;
    const float* row = scores + (size_t)q * lds;
    for (int i = threadIdx.x; i < P; i += 256) {
        bins[i] = counts[(size_t)q * ldp + i];
        slot[i] = __builtin_bit_cast(int, probe_scores[(size_t)q * ldp + i]);
    }
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    // suffix sums, 256 positions at a time from the top: S[i] = sum_{m >= i} bins[m]
    __shared__ int part[256];
    for (int top = ((P + 255) / 256) * 256; top > 0; top -= 256) {
        const int i = top - 256 + (int)threadIdx.x;
        const int x = i < P ? bins[i] : 0;
        part[threadIdx.x] = x;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {   // inclusive suffix scan (Hillis-Steele)
            const int y = (int)threadIdx.x + d < 256 ? part[threadIdx.x + d] : 0;
            __syncthreads();
            part[threadIdx.x] += y;
            __syncthreads();
        }
        const int c = carry;
        __syncthreads();
        if (i < P) bins[i] = part[threadIdx.x] + c;
        if (threadIdx.x == 0) carry = c + part[0];
        __syncthreads();
    }
    for (int i = threadIdx.x; i < P; i += 256) {
        const int sl = slot[i];
        const int idx = probe_idx[(size_t)q * ldp + sl];
        counts[(size_t)q * ldp + sl] = idx >= 0 ? bins[i] : 0;
        probe_scores[(size_t)q * ldp + sl] = idx >= 0 ? row[idx] : 0.f;
    }
}

int rank_counts(const float* scores, int lds, int Q, int N, const int* probe_idx, int P, int* counts,
                float* probe_scores, hipStream_t stream) {
    if (Q <= 0 || P <= 0 || N <= 0) return DIR_OK;
    if (lds < N) return fail(DIR_ERR_INVALID, "rank_counts: lds < N");
    DIR_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)Q * P * sizeof(int), stream));
    static std::atomic<uint64_t> attr_sort{0}, attr_hist{0}, attr_fin{0};
    DIR_HIP_CHECK(ensure_dynamic_lds((const void*)rank_sort_kernel, kMaxProbes * 12, attr_sort));
    DIR_HIP_CHECK(ensure_dynamic_lds((const void*)rank_hist_kernel, kMaxProbes * 12, attr_hist));
    DIR_HIP_CHECK(ensure_dynamic_lds((const void*)rank_finalize_kernel, kMaxProbes * 8, attr_fin));
    const long chunks = ((long)N + kRankChunk - 1) / kRankChunk;
    for (int p0 = 0; p0 < P; p0 += kMaxProbes) {   // 4096 probes per query per pass, column slices of [Q][P]
        const int np = P - p0 < kMaxProbes ? P - p0 : kMaxProbes;
        int p2 = 2;
        while (p2 < np) p2 <<= 1;
        int* perm = (int*)(probe_scores + p0);     // the slice of probe_scores holds the permutation until the last kernel
        hipLaunchKernelGGL(rank_sort_kernel, dim3(Q), dim3(256), (size_t)p2 * 12, stream, scores, lds, probe_idx + p0, np,
                           p2, P, perm);
        hipLaunchKernelGGL(rank_hist_kernel, dim3((unsigned)chunks, Q), dim3(256), (size_t)p2 * 12, stream, scores, lds, N,
                           probe_idx + p0, np, p2, P, perm, counts + p0);
        hipLaunchKernelGGL(rank_finalize_kernel, dim3(Q), dim3(256), (size_t)np * 8, stream, scores, lds, probe_idx + p0, np,
                           P, counts + p0, probe_scores + p0);
        DIR_HIP_CHECK(hipGetLastError());
    }
    return DIR_OK;
}

// ---- N3: alpha query expansion / database augmentation (dirtorch/test_dir.py:24-44) ---------------
// out[i] = normalize( (descs[i] + sum_{j in top-k of sim[i]} sim[i][j]^alpha * db[j]) / (k + 1) ),
// sim = descs . db^T (computed by the fp32 MFMA GEMM into `sim`), the diagonal zeroed when the set is
// expanded against itself (test_dir.py:33-34).  One workgroup per row: k selection rounds over the
// row (each finds the best item ranking after the previous pick - value descending, index descending
// on ties, the order of np.argsort(...)[::-1]; the reference's np.argpartition leaves the choice among
// tied boundary values unspecified), then the weighted sum and the L2 norm.  Picks are buffered 256 at a time
// (any k <= m, like the reference); the running sum lives in the output row between batches.  A row with
// non-finite similarities (a NaN descriptor from --load-feats, say) can run out of selectable items: the pick is
// then (row 0, weight NaN) and the output row is NaN, which is what the reference's arithmetic propagates.
constexpr int kMaxExpandK = 256;

__global__ void __launch_bounds__(256) expand_rows_kernel(const float* __restrict__ descs,
                                                         const float* __restrict__ db,
                                                         float* __restrict__ sim, int row0, int m, int D,
                                                         int k, float alpha, int self_set,
                                                         float* __restrict__ out) {
    __shared__ float s_val[256];
    __shared__ int s_idx[256];
    __shared__ float s_pick_w[kMaxExpandK];
    __shared__ int s_pick_j[kMaxExpandK];
    __shared__ float s_red[256];
    const int r = blockIdx.x;          // row within this chunk
    const int i = row0 + r;            // row of descs / out
    float* row = sim + (size_t)r * m;
    const int tid = threadIdx.x;
    if (self_set && tid == 0 && i < m) row[i] = 0.f;   // sim[np.diag_indices(n)] = 0
    __syncthreads();
    float pv = INFINITY;
    int pj = 0x7fffffff;
    const int ialpha = (int)alpha;
    const bool int_alpha = (float)ialpha == alpha && ialpha >= 0 && ialpha <= 64;
    for (int d = tid; d < D; d += 256) out[(size_t)i * D + d] = descs[(size_t)i * D + d];
    for (int round = 0; round < k; ++round) {
        float bv = -INFINITY;
        int bj = -1;
        for (int j = tid; j < m; j += 256) {
            const float v = row[j];
            const bool after = (v < pv) || (v == pv && j < pj);
            if (after && (v > bv || (v == bv && j > bj))) {
                bv = v;
                bj = j;
            }
        }
        s_val[tid] = bv;
        s_idx[tid] = bj;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                const float ov = s_val[tid + s];
                const int oj = s_idx[tid + s];
                if (oj >= 0 && (s_idx[tid] < 0 || ov > s_val[tid] || (ov == s_val[tid] && oj > s_idx[tid]))) {
                    s_val[tid] = ov;
                    s_idx[tid] = oj;
                }
            }
            __syncthreads();
        }
        pv = s_val[0];
        pj = s_idx[0];
        if (tid == 0) {
            float w;
            if (int_alpha) {           // sim ** alpha with the CLI's integer alpha (test_dir.py:212-213)
                w = 1.f;
                for (int e = 0; e < ialpha; ++e) w *= pv;
            } else {
                w = powf(pv, alpha);
            }
            s_pick_w[round % kMaxExpandK] = pj < 0 ? NAN : w;
            s_pick_j[round % kMaxExpandK] = pj < 0 ? 0 : pj;
        }
        __syncthreads();
        const int filled = round % kMaxExpandK + 1;
        if (filled == kMaxExpandK || round == k - 1) {   // flush this batch of picks into the running sum
            for (int d = tid; d < D; d += 256) {
                float acc = out[(size_t)i * D + d];
                for (int t = 0; t < filled; ++t) acc += db[(size_t)s_pick_j[t] * D + d] * s_pick_w[t];
                out[(size_t)i * D + d] = acc;
            }
            __syncthreads();
        }
    }
    // weighted mean over the k + 1 rows, then the L2 norm (test_dir.py:38-42)
    const float inv = 1.f / (float)(k + 1);
    float ss = 0.f;
    for (int d = tid; d < D; d += 256) {
        const float acc = out[(size_t)i * D + d] * inv;
        out[(size_t)i * D + d] = acc;
        ss += acc * acc;
    }
    s_red[tid] = ss;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) s_red[tid] += s_red[tid + s];
        __syncthreads();
    }
    const float nrm = sqrtf(s_red[0]);
    for (int d = tid; d < D; d += 256) out[(size_t)i * D + d] /= nrm;
}

int expand_descriptors(const float* descs, int n, const float* db, int m, int D, int k, float alpha,
                       int self_set, float* out, float* sim, size_t sim_bytes, hipStream_t stream) {
    if (n <= 0) return DIR_OK;
    if (k < 0 || alpha < 0.f) return fail(DIR_ERR_INVALID, "expand_descriptors: k and alpha must be non-negative");
    if (k > m) return fail(DIR_ERR_INVALID, "expand_descriptors: k exceeds the number of candidate rows");
    if (self_set && m != n) return fail(DIR_ERR_INVALID, "expand_descriptors: self expansion needs db == descs");
    const size_t rows_fit = sim_bytes / ((size_t)m * sizeof(float));
    if (rows_fit == 0) return fail(DIR_ERR_WORKSPACE, "expand_descriptors: scratch smaller than one score row");
    for (int r0 = 0; r0 < n; r0 += (int)std::min<size_t>(rows_fit, 1 << 30)) {
        const int rows = (int)std::min<size_t>(rows_fit, (size_t)(n - r0));
        // sim[r][j] = <descs[r0 + r], db[j]>  (P = db: the long operand; exact fp32 MFMA)
        int rc = gemm_nt_f32(db, D, descs + (size_t)r0 * D, D, sim, m, m, rows, D, nullptr, nullptr, nullptr, stream);
        if (rc != DIR_OK) return rc;
        hipLaunchKernelGGL(expand_rows_kernel, dim3(rows), dim3(256), 0, stream, descs, db, sim, r0, m, D, k,
                           alpha, self_set, out);
        DIR_HIP_CHECK(hipGetLastError());
    }
    return DIR_OK;
}

// ---- device-side AP of the revisited protocol (generic.py:196-224 + evaluation.py:46-82) -----------
// Input: the dense counts / scores of every listed image of every query (rank_counts above, union
// list `probe_idx`), and per (query, mode) two index lists INTO that union list: the positives and the
// junk of the mode (host-prepared once per dataset: duplicates removed, an image that is both is junk).
//   rank(p)  = count[p] - #{junk j ranking before p}        position among the kept (non-junk) images
//   i(p)     = #{positives ranking before p}                position among the sorted positive ranks
//   AP       = sum_i ((i/rank or 1) + (i+1)/(rank+1)) / (2 n)   added in order of i, in fp64, exactly the
//              loop of compute_average_precision
// One workgroup per (query, mode); AP = -1 when the mode has no positive (generic.py:217-218).
__global__ void __launch_bounds__(256) revisitop_ap_kernel(const int* __restrict__ probe_idx, int P,
                                                          const int* __restrict__ counts,
                                                          const float* __restrict__ pscores,
                                                          const int* __restrict__ pos_off,
                                                          const int* __restrict__ pos_list,
                                                          const int* __restrict__ junk_off,
                                                          const int* __restrict__ junk_list, int modes,
                                                          double* __restrict__ terms,
                                                          double* __restrict__ ap_out) {
    const int q = blockIdx.x, mode = blockIdx.y, qm = q * modes + mode;
    const int p0 = pos_off[qm], np_ = pos_off[qm + 1] - p0;
    const int j0 = junk_off[qm], nj = junk_off[qm + 1] - j0;
    if (np_ == 0) {
        if (threadIdx.x == 0) ap_out[qm] = -1.0;
        return;
    }
    const int* pidx = probe_idx + (size_t)q * P;
    const int* cnt = counts + (size_t)q * P;
    const float* psc = pscores + (size_t)q * P;
    double* tq = terms + p0;   // one slot per positive of this (query, mode)
    for (int a = threadIdx.x; a < np_; a += 256) {
        const int ka = pos_list[p0 + a];
        const float sa = psc[ka];
        const int ia = pidx[ka];
        int before_junk = 0, before_pos = 0;
        for (int b = 0; b < nj; ++b) {
            const int kb = junk_list[j0 + b];
            const float sb = psc[kb];
            const int ib = pidx[kb];
            before_junk += (sb > sa) || (sb == sa && ib > ia);
        }
        for (int b = 0; b < np_; ++b) {
            const int kb = pos_list[p0 + b];
            const float sb = psc[kb];
            const int ib = pidx[kb];
            before_pos += (sb > sa) || (sb == sa && ib > ia);
        }
        const long rank = (long)cnt[ka] - before_junk;
        const double i = (double)before_pos;
        const double left = rank == 0 ? 1.0 : i / (double)rank;
        const double right = (i + 1.0) / ((double)rank + 1.0);
        const double step = 1.0 / (double)np_;
        tq[before_pos] = (left + right) * step / 2.0;   // slot = position in the sorted order
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ap = 0.0;
        for (int a = 0; a < np_; ++a) ap += tq[a];      // the reference's summation order
        ap_out[qm] = ap;
    }
}

int revisitop_ap(const int* probe_idx, int Q, int P, const int* counts, const float* pscores, const int* pos_off,
                 const int* pos_list, const int* junk_off, const int* junk_list, int modes, double* terms,
                 double* ap_out, hipStream_t stream) {
    if (Q <= 0 || modes <= 0) return DIR_OK;
    hipLaunchKernelGGL(revisitop_ap_kernel, dim3(Q, modes), dim3(256), 0, stream, probe_idx, P, counts, pscores,
                       pos_off, pos_list, junk_off, junk_list, modes, terms, ap_out);
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

}  // namespace dir
