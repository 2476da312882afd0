// conv_igemm.h — host-side contract of the implicit-GEMM convolution family (conv_igemm.hip).
#pragma once
#include "dir_common.h"

namespace dir {

struct ConvArgs {
    const uint16_t* x;     // NHWC [B,H,W,Cin]
    const uint16_t* w;     // [Cout][R][S][Cin]
    const float* bias;     // [Cout]  (folded BatchNorm shift)
    const uint16_t* res;   // NHWC [B,OH,OW,Cout] or nullptr
    uint16_t* y;           // NHWC [B,OH,OW,Cout]
    int B, H, W, Cin, OH, OW, Cout;
    int R, S, stride, pad, relu;
    int M;     // B*OH*OW
    int Ktot;  // R*S*Cin
    int T;     // Ktot / 64
    // split-K (small M: too few output tiles to fill the chip).  ksplit > 1: workgroup (tile, z)
    // accumulates K-steps [z*T'/ksplit, (z+1)*T'/ksplit) and stores raw fp32 sums to
    // partial[z][M][Cout]; conv_splitk_finalize adds them in z order with bias / residual / ReLU.
    int ksplit;      // 0 or 1 = off
    float* partial;
    // fused bottleneck seam (conv_c3c1.hip): the NEXT block's conv1 applied to this conv's output tile
    const uint16_t* w2;    // [Cout2][Cout]
    const float* bias2;    // [Cout2]
    uint16_t* y2;          // NHWC [B,OH,OW,Cout2]
    int Cout2, relu2;
    // ... and, for the first block of a stage, the block input as a second K source (the downsample conv
    // folded into this GEMM): x2 [M][Cin2], its weights appended to w along K, its bias added to bias
    const uint16_t* x2;
    int Cin2;
    // ... or (two-source GEMM, conv_persist.hip DUAL) any 1x1 downsample: x2 is NHWC [B,H2,W2,Cin2], output
    // pixel (b, oh, ow) reads x2 pixel (b, oh*stride2, ow*stride2)
    int H2, W2, stride2;
    uint32_t x2_bytes;
    int* ovf;              // fp16 overflow word (dir_common.h Ovf), or nullptr
    // paired-fp16 form (conv_pair.hip, DIR_FP16P): every operand is a PAIR of fp16 planes, value = hi + lo with
    // lo = fp16(v - hi) (~22 significant bits); x / w / res / y above are the hi planes, these the lo planes of the
    // same layout.  w_lo is required there; x_lo / res_lo / y_lo may be null (single-plane operand / output).
    const uint16_t* x_lo;
    const uint16_t* x2_lo;   // ... and of the second K source (x2) in the two-source form
    const uint16_t* w_lo;
    const uint16_t* w_pw;    // conv_patchw.hip (loader / consumer form): w as its LDS stage images (conv_patch3x3w_pack), or null (gathered from w)
    const uint16_t* w_s2;    // conv_patchs2.hip: w in fragment order (conv_patch3x3s2_pack), or null (the launcher packs into scratch)
    const uint16_t* w2_lo;   // the fused seam (conv_c3c1.hip, WP1): lo plane of the following conv1's weights
    const uint16_t* res_lo;
    uint16_t* y_lo;
    // filled by the launcher
    int tiles_m, tiles_n;
    uint32_t x_bytes, w_bytes;               // buffer-descriptor extents (bounds-checked DMA)
    int rev_m;                               // persistent 1x1: walk the pixel tiles from the LAST one (read what the
                                             // producer wrote most recently first: it is still in the 256 MB Infinity Cache)
    int no_xcd_map;                          // persistent kernels: 1 = tiles follow blockIdx (A/B; default 0 = XCD-aware)
    int flat;                                // 1x1, stride 1, no padding: pixel m reads pixel m
    uint32_t div_ohw_mul, div_ohw_shr;       // exact n / (OH*OW) and n / OW for n < 2^31
    uint32_t div_ow_mul, div_ow_shr;
};

typedef hipError_t (*ConvLaunchFn)(const ConvArgs&, hipStream_t);

struct ConvVariant {
    const char* name;
    int BM, BN, threads, stages, BK;
    ConvLaunchFn launch[2];    // [dtype]
    ConvLaunchFn launch16[2];  // Cin == 16 stem instantiation, or nullptr
    int kind;                  // 0 = implicit GEMM (conv_igemm.hip), 1 = LDS-patch 3x3 (conv_patch.hip),
                               // 2 = persistent 256x256 1x1 (conv_persist.hip),
                               // 3 = register-stationary weights 1x1 (conv_wreg.hip),
                               // 4 = kind 2 with the pixel operand three K-steps deep (conv_persist.hip, XDEEP),
                               // 5 = LDS-patch 3x3 for wide layers, one 64-channel plane at a time (conv_patch.hip)
                               // 6 = LDS-patch 3x3, 512 pixels x 128 channels, double-buffered 32-channel planes (conv_patchw.hip)
                               // 7 = persistent 128x256 1x1 without a residual, one K ring over all tiles of a workgroup (conv_ring.hip)
                               // 8 = 64 -> 64 channel 3x3 with the filter resident in LDS, loader / consumer waves (conv_patchlc.hip)
                               // 9 = two-source 1x1 (launch_dual only) with register-stationary weights (conv_wregd.hip)
                               // 11 = 64 x 64 tiles for small maps: loader / consumer waves on LDS counters (conv_small.hip; `stages` = ring depth)
                               // 10 = kind 4's ring with loader / consumer wave roles, 1x1 without a residual and its two-source form (conv_persistlc.hip)
    ConvLaunchFn launch_sk[2]; // split-K instantiation (ConvArgs::ksplit > 1), or nullptr
    ConvLaunchFn launch_dual[2]; // two-source K instantiation (ConvArgs::x2: conv3 + downsample in one GEMM), or nullptr
};

bool conv1x1_persist_admissible(const ConvArgs& a);
hipError_t conv1x1_persist_launch(const ConvArgs& a, int dtype, hipStream_t stream, bool xdeep = false);
hipError_t conv1x1_persist_dual_bf16(const ConvArgs& a, hipStream_t stream);   // conv3 + downsample, two K sources
hipError_t conv1x1_persist_dual_fp16(const ConvArgs& a, hipStream_t stream);
bool conv_patch64_lc_admissible(const ConvArgs& a);
hipError_t conv_patch64_lc_launch(const ConvArgs& a, int dtype, hipStream_t stream);
bool conv1x1_ring_admissible(const ConvArgs& a);
hipError_t conv1x1_ring_launch(const ConvArgs& a, int dtype, hipStream_t stream);
bool conv1x1_wreg_admissible(const ConvArgs& a);
bool conv_small_admissible(const ConvArgs& a);      // small maps: 64 x 64 tiles, four consumer + four loader waves, no workgroup barrier (conv_small.hip)
hipError_t conv_small_launch(const ConvArgs& a, int dtype, int nst, hipStream_t stream);
bool conv1x1_lc_admissible(const ConvArgs& a);      // the persistent 256 x 256 ring with loader / consumer roles (conv_persistlc.hip)
hipError_t conv1x1_lc_launch(const ConvArgs& a, int dtype, hipStream_t stream);
hipError_t conv1x1_lc_dual_bf16(const ConvArgs& a, hipStream_t stream);
hipError_t conv1x1_lc_dual_fp16(const ConvArgs& a, hipStream_t stream);
bool conv1x1_wregd_admissible(const ConvArgs& a);   // two-source form, K = 128 + 256 (layer2's first block)
hipError_t conv1x1_wregd_bf16(const ConvArgs& a, hipStream_t stream);
hipError_t conv1x1_wregd_fp16(const ConvArgs& a, hipStream_t stream);
hipError_t conv1x1_wreg_launch(const ConvArgs& a, int dtype, hipStream_t stream);
// the layer3 seam (planes 256): weights streamed from L2 through an LDS ring by loader waves (conv_seam3.hip); reached
// through conv_c3c1_admissible / conv_c3c1_launch like the register-stationary forms
bool conv_seam3_admissible(const ConvArgs& a);
hipError_t conv_seam3_launch(const ConvArgs& a, int dtype, hipStream_t stream);
bool conv_c3c1_admissible(const ConvArgs& a);
bool conv_c3c1ds_lc_admissible(const ConvArgs& a);   // the DS seam, loader / consumer form (conv_c3c1lc.hip)
hipError_t conv_c3c1ds_lc_launch(const ConvArgs& a, int dtype, hipStream_t stream);
hipError_t conv_c3c1_launch(const ConvArgs& a, int dtype, hipStream_t stream);
bool conv_patch3x3s_admissible(const ConvArgs& a);
hipError_t conv_patch3x3s_launch(const ConvArgs& a, int dtype, hipStream_t stream);
bool conv_patch3x3w_admissible(const ConvArgs& a);
hipError_t conv_patch3x3w_launch(const ConvArgs& a, int dtype, hipStream_t stream);
hipError_t conv_patch3x3w_pack(const uint16_t* w, uint16_t* out, int Cout, int Cin, hipStream_t stream);   // same size as w
bool conv_patch3x3s2_admissible(const ConvArgs& a);   // conv_patchs2.hip: 3x3 stride 2
hipError_t conv_patch3x3s2_launch(const ConvArgs& a, int dtype, hipStream_t stream);
hipError_t conv_patch3x3s2_pack(const uint16_t* w, uint16_t* out, int Cout, int Cin, hipStream_t stream);   // same size as w
bool conv_patch3x3_admissible(const ConvArgs& a);
hipError_t conv_patch3x3_launch(const ConvArgs& a, int dtype, hipStream_t stream);

// Paired-fp16 convolution (conv_pair.hip): three fp16 MFMAs per product term (wh.xh + wh.xl + wl.xh) into one fp32
// accumulator; any R x S <= 4 x 4, stride, padding; Cin % 32 == 0, Cout % 64 == 0.  Returns a dir_status.
int conv_pair_launch(const ConvArgs& a, hipStream_t stream);
const char* conv_pair_variant_name(const ConvArgs& a);

int conv_variant_count();
const ConvVariant& conv_variant(int i);
bool conv_variant_admissible(int v, const ConvArgs& a);
int conv_pick_variant(const ConvArgs& a);
// Variant for the two-source form (a.x2 set), or -1 when none fits the shape.
int conv_pick_dual_variant(const ConvArgs& a, bool any_size = false);
int conv_launch(const ConvArgs& a, int dtype, int variant, hipStream_t stream);
// Split factor for variant `v` on problem `a` (1 = none) and the fp32 scratch it needs.
int conv_splitk_factor(int v, const ConvArgs& a);
size_t conv_splitk_bytes(const ConvArgs& a, int ksplit);
constexpr size_t kSplitKMaxBytes = 64u << 20;   // scratch the engine reserves for the partial sums
int conv_launch_naive(const ConvArgs& a, int dtype, hipStream_t stream);


}  // namespace dir
