// ranking.hip — device-side ranking for the revisited Oxford/Paris AP protocol.
//
// The reference downloads the Q x N score matrix and runs np.argsort over every row three times
// (dirtorch/datasets/generic.py:196-224); at N = 10^6 that is 280 MB of D2H and 210 million-element
// sorts.  AP only needs the rank of each positive among the non-junk images, i.e.
//     rank(p) = #{ j : j ranks before p } - #{ junk j : j ranks before p },
// where "j ranks before p" is (s_j > s_p) or (s_j == s_p and j > p) - the order of
// np.argsort(scores)[::-1] with ties resolved by descending index.  The first term is a dense count
// over the row (this kernel, HBM-bound: each score read once per 1024 probes); the second involves
// only the few hundred listed images of the query and is finished on the host from the probe scores
// this kernel also returns.
#include "dir_common.h"
#include "pointwise.h"

#include <algorithm>

namespace dir {

constexpr int kRankChunk = 4096;  // scores staged per workgroup (16 KiB of LDS)
constexpr int kMaxProbes = 1024;  // probes per query per launch (4 per lane)

__global__ void __launch_bounds__(256) rank_counts_kernel(const float* __restrict__ scores, int lds,
                                                         int N, const int* __restrict__ probe_idx,
                                                         int P, int ldp, int* __restrict__ counts,
                                                         float* __restrict__ probe_scores) {
    __shared__ __attribute__((aligned(16))) float tile[kRankChunk];
    const int q = blockIdx.y;
    const int j0 = blockIdx.x * kRankChunk;
    const float* row = scores + (size_t)q * lds;
    const int n = min(kRankChunk, N - j0);
    for (int i = threadIdx.x; i < kRankChunk; i += 256)
        tile[i] = i < n ? row[j0 + i] : -INFINITY;  // -inf never ranks before a finite probe

    int pidx[4];
    float ps[4];
    int cnt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = threadIdx.x + 256 * u;
        pidx[u] = p < P ? probe_idx[(size_t)q * ldp + p] : -1;
        ps[u] = pidx[u] >= 0 ? row[pidx[u]] : INFINITY;
        cnt[u] = 0;
    }
    __syncthreads();
    const int nu = (P + 255) / 256;  // probe slots in use (uniform)
    for (int i = 0; i < kRankChunk; i += 4) {
        const f32x4_t s4 = *(const f32x4_t*)(tile + i);  // same address for every lane: broadcast
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = j0 + i + e;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < nu) cnt[u] += (s4[e] > ps[u]) || (s4[e] == ps[u] && j > pidx[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = threadIdx.x + 256 * u;
        if (p < P && pidx[u] >= 0) {
            if (cnt[u]) atomicAdd(counts + (size_t)q * ldp + p, cnt[u]);
            if (blockIdx.x == 0) probe_scores[(size_t)q * ldp + p] = ps[u];
        }
    }
}

int rank_counts(const float* scores, int lds, int Q, int N, const int* probe_idx, int P, int* counts,
                float* probe_scores, hipStream_t stream) {
    if (Q <= 0 || P <= 0 || N <= 0) return DIR_OK;
    if (lds < N) return fail(DIR_ERR_INVALID, "rank_counts: lds < N");
    DIR_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)Q * P * sizeof(int), stream));
    const dim3 grid((N + kRankChunk - 1) / kRankChunk, Q);
    for (int p0 = 0; p0 < P; p0 += kMaxProbes) {   // 1024 probes per query per launch, column slices of [Q][P]
        const int np = P - p0 < kMaxProbes ? P - p0 : kMaxProbes;
        hipLaunchKernelGGL(rank_counts_kernel, grid, dim3(256), 0, stream, scores, lds, N, probe_idx + p0, np, P,
                           counts + p0, probe_scores + p0);
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
