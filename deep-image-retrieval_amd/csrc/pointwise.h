// pointwise.h — launchers of the HBM-bound kernels (pointwise.hip) and the fp32 NT GEMM (gemm_f32.hip).
#pragma once
#include <vector>

#include "dir_common.h"

namespace dir {

int prep_input(const void* img, int fmt, const float* mean3, const float* std3, void* out, int B,
               int H, int W, int dtype, hipStream_t stream);
int maxpool_3x3s2(const void* x, void* y, int B, int H, int W, int C, int dtype, hipStream_t stream);
// out row b starts at out + b * ldo (the FPN heads pool two maps into one concatenated row)
int global_pool(const void* x, float* out, int ldo, int B, int H, int W, int C, int pooling, float p,
                float eps, float center_bias, int dtype, hipStream_t stream);
int upsample_add(const void* x, const void* low, void* y, int B, int H, int W, int h, int w, int C,
                 int dtype, hipStream_t stream, int* ovf = nullptr);
int l2norm_rows(float* x, int rows, int cols, float eps, hipStream_t stream);
size_t resize_workspace_bytes(int B, int H, int W, int OH, int OW);
int resize_bilinear_u8(const uint8_t* src, uint8_t* dst, int B, int H, int W, int OH, int OW, void* ws,
                       size_t ws_bytes, hipStream_t stream);
int multiscale_pool(const float* x, float* out, int S, int N, int D, int mode, float gemp,
                    hipStream_t stream);
int stem_pool_launch(const void* s2d, const void* w, const float* bias, void* y, int B, int H2, int W2,
                     int OH, int OW, int dtype, hipStream_t stream, int* ovf = nullptr);
// the paired-fp16 forms of prep_input and of the stem (conv_pair.hip, DIR_FP16P): every tensor is two fp16 planes
int prep_input_pair(const void* img, int fmt, const float* mean3, const float* std3, void* out_hi, void* out_lo, int B,
                    int H, int W, hipStream_t stream);
int stem_pool_pair_launch(const void* s2d_hi, const void* s2d_lo, const void* w_hi, const void* w_lo, const float* bias,
                          void* y_hi, void* y_lo, int B, int H2, int W2, int OH, int OW, hipStream_t stream,
                          int* ovf = nullptr);
// DIR_FP16P on the raw uint8 feed (stem_u8.hip): one exact plane of u / 256, Normalize folded into the filter pair / bias / border table
int fold_stem_u8(const float* w, const float* scale, const float* bias, const float* mean3, const float* std3,
                 std::vector<uint16_t>& hi, std::vector<uint16_t>& lo, std::vector<float>& b2, std::vector<float>& corr);
int prep_input_u8(const void* img, void* out, int B, int H, int W, hipStream_t stream);
// img: the uint8 NHWC image (RAW form when stem_pool_u8_raw_ok: no prep launch, s2d may be null) or null (s2d = prep_input_u8's plane)
bool stem_pool_u8_raw_ok(const void* img, int B, int H, int W);
int stem_pool_u8_launch(const void* img, const void* s2d, const void* w_hi, const void* w_lo, const float* bias, const float* corr,
                        void* y_hi, void* y_lo, int B, int H, int W, hipStream_t stream, int* ovf = nullptr, int seg_tiles = 0);
// ... and the generic paired stem on the same kernel structure (stem_u8.hip XPAIR): what stem_pool_pair_launch runs by default
bool stem_pool_pair_raw_ok(const void* img_f32, int B, int H, int W);
int stem_pool_pair_walk_launch(const void* s2d_hi, const void* s2d_lo, const void* w_hi, const void* w_lo, const float* bias,
                               void* y_hi, void* y_lo, int B, int H2, int W2, int OH, int OW, hipStream_t stream, int* ovf = nullptr,
                               const void* img_f32 = nullptr, int H = 0, int W = 0);
int rank_counts(const float* scores, int lds, int Q, int N, const int* probe_idx, int P, int* counts,
                float* probe_scores, hipStream_t stream);
int revisitop_ap(const int* probe_idx, int Q, int P, const int* counts, const float* pscores, const int* pos_off,
                 const int* pos_list, const int* junk_off, const int* junk_list, int modes, double* terms,
                 double* ap_out, hipStream_t stream);
int expand_descriptors(const float* descs, int n, const float* db, int m, int D, int k, float alpha,
                       int self_set, float* out, float* sim, size_t sim_bytes, hipStream_t stream);
// scratch (optional): fp32 workspace for the split-K partial sums of shapes with few output tiles; without one
// the partials live in stream-ordered memory (hipMallocAsync)
constexpr size_t kGemmSplitKMaxBytes = 64u << 20;
int gemm_splitk_factor(int NP, int NQ, int K);
int gemm_nt_f32(const float* P, int ldp, const float* Q, int ldq, float* out, int ldo, int NP,
                int NQ, int K, const float* qsub, const float* bias, const float* alpha,
                hipStream_t stream, float* scratch = nullptr, size_t scratch_bytes = 0);
// large-database similarity on the bf16 matrix cores at fp32 accuracy (sim_split.hip)
size_t similarity_split_workspace_bytes(int NQ, int K);
bool similarity_split_admissible(const float* P, int ldp, const float* Q, int ldq, int NP, int NQ, int K);
// unit_range: operands known to lie in (-64, 64) (L2-normalised descriptors): two fp16 planes instead of three bf16 ones
int similarity_split(const float* P, int ldp, const float* Q, int ldq, float* out, int ldo, int NP, int NQ, int K,
                     void* workspace, size_t workspace_bytes, hipStream_t stream, bool unit_range = false);

// PCA whitening of a large descriptor set on two fp16 planes per operand (sim_split.hip): out[n][j] = alpha[j] <X[n] - mean, C[j]>
size_t whiten_split_workspace_bytes(int v, int K);
bool whiten_split_admissible(const float* X, int ldx, int N, int K, int v);
int whiten_split(const float* X, int ldx, int N, const float* comps, int ldc, int v, int K, const float* mean, const float* alpha,
                 float* out, int ldo, void* workspace, size_t workspace_bytes, hipStream_t stream);

}  // namespace dir
