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

namespace dir {

constexpr int kRankChunk = 4096;  // scores staged per workgroup (16 KiB of LDS)
constexpr int kMaxProbes = 1024;  // probes per query per launch (4 per lane)

__global__ void __launch_bounds__(256) rank_counts_kernel(const float* __restrict__ scores, int lds,
                                                         int N, const int* __restrict__ probe_idx,
                                                         int P, int* __restrict__ counts,
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
        pidx[u] = p < P ? probe_idx[(size_t)q * P + p] : -1;
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
            if (cnt[u]) atomicAdd(counts + (size_t)q * P + p, cnt[u]);
            if (blockIdx.x == 0) probe_scores[(size_t)q * P + p] = ps[u];
        }
    }
}

int rank_counts(const float* scores, int lds, int Q, int N, const int* probe_idx, int P, int* counts,
                float* probe_scores, hipStream_t stream) {
    if (Q <= 0 || P <= 0 || N <= 0) return DIR_OK;
    if (P > kMaxProbes)
        return fail(DIR_ERR_INVALID, "rank_counts: more than 1024 probes per query; split the call");
    if (lds < N) return fail(DIR_ERR_INVALID, "rank_counts: lds < N");
    DIR_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)Q * P * sizeof(int), stream));
    const dim3 grid((N + kRankChunk - 1) / kRankChunk, Q);
    hipLaunchKernelGGL(rank_counts_kernel, grid, dim3(256), 0, stream, scores, lds, N, probe_idx, P,
                       counts, probe_scores);
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

}  // namespace dir
