// engine.h — model graph, weight packing and the forward pass of ResNet_RMAC on one MI355X.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "conv_f32.h"
#include "conv_igemm.h"
#include "pointwise.h"

namespace dir {

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

struct ConvLayer {
    std::string name;     // "layer3.4.conv2", "conv1", "layer2.0.downsample"
    std::string wkey;     // state-dict key of the conv weight
    std::string bnprefix; // state-dict prefix of its BatchNorm ("layer3.4.bn2")
    int Cin = 0, Cout = 0, R = 1, S = 1, stride = 1, pad = 0;
    bool relu = false;
    bool stem = false;    // packed as the 4x4 s1 space-to-depth form (Cin 16)
    uint16_t* d_w = nullptr;
    float* d_wf = nullptr;   // DIR_F32 (strict path, conv_f32.hip): the same layout in fp32; d_w stays null
    uint16_t* d_w_lo = nullptr;  // DIR_FP16P, the paired layers (stem, layer1; conv_pair.hip): fp16(w - fp16(w)), same layout
    uint16_t* d_w_pw = nullptr;  // 3x3 stride-1 layers over >= 64 channels (conv_patchw.hip): d_w as that kernel's LDS stage images
    uint16_t* d_w_s2 = nullptr;  // 3x3 stride-2 layers (conv_patchs2.hip): d_w in that kernel's fragment order
    float* d_bias = nullptr;
    // conv3 of a stage's first block whose downsample qualifies (conv_c3c1.hip, DS form): this conv's
    // weights with the downsample's appended along K, and the sum of the two folded-BN biases
    uint16_t* d_w_ds = nullptr;
    uint16_t* d_w_ds_lo = nullptr;   // ... its lo plane (DIR_FP16P paired head, conv_pair.hip's two-source form)
    float* d_bias_ds = nullptr;
    std::vector<uint16_t> h_w;   // host copies, alive during finalize() only
    std::vector<uint16_t> h_w_lo;
    std::vector<float> h_bias;
    std::map<long, int> tuned;  // M -> variant index chosen by autotune
};

struct BlockDef {
    int conv1 = -1, conv2 = -1, conv3 = -1, down = -1;
    int stride = 1;
};

struct ProfSlot {
    hipEvent_t start, stop;
    std::string name, kernel;
    double flops, bytes;
};

struct Plan {  // byte offsets into the caller's workspace for one (B, H, W)
    size_t s2d, stem, bufA, bufB, t1, t2, ds, x4, splitk, pooled, fcout, total;
    size_t lo_s2d, lo_stem, lo_t1, lo_t2, lo_ds;   // DIR_FP16P: lo planes of the paired head's tensors
    int H2, W2, OH1, OW1, PH, PW;
};

}  // namespace dir

struct dir_engine {
    dir_model_desc desc;
    int device = 0;
    int dtype = DIR_BF16;
    // DIR_FP16P: how many leading blocks (all of layer1 by default) run on fp16 pairs (conv_pair.hip); the stem always does
    int pair_blocks = 0;
    // ... and which tensors there are pairs.  false (bottleneck nets, the default): the image, the stem (weights and
    // output) and the 1x1 WEIGHTS of those blocks - lo planes that sit in registers of the HBM-bound seam kernels and cost
    // MFMAs only; activations and the 3x3 weights stay single fp16 planes (1 - cos 4.2e-5 / 3.0e-5 at config A / B).
    // true (DIRTORCH_AMD_PAIR_ACTS=1 at finalize, and always for BasicBlock nets, whose layer1 has no 1x1): every
    // weight and every tensor INSIDE those blocks (t1, t2, the downsample branch) is a pair too (1.7e-5 / 1.5e-5).
    bool pair_acts = false;
    // the 16-bit kernels' dtype: DIR_FP16P stores and multiplies fp16 everywhere, pairs of them in the head
    int kdtype() const { return dtype == DIR_FP16P ? DIR_FP16 : dtype; }
    bool finalized = false;
    std::map<std::string, dir::HostTensor> state;
    std::vector<dir::ConvLayer> convs;
    std::vector<dir::BlockDef> blocks;
    int feat_dim = 0;  // trunk channels (512 * expansion)
    int x4_dim = 0;    // layer3 channels (256 * expansion), FPN heads only
    int head_dim = 0;  // width of the pooled vector that feeds the FC (feat_dim, or x4_dim + feat_dim)
    int x4_block = -1; // index of the last layer3 block (its output is kept for the FPN heads)
    int conv1x5 = -1, conv3c4 = -1;  // lateral convs of DIR_HEAD_FPN (no BatchNorm, no bias)
    float gem_p = 3.f;   // adpool.p (RMAC) or adpoolx5.p (FPN)
    float gem_p4 = 3.f;  // adpoolc4.p
    float* d_fc_w = nullptr;
    float* d_fc_b = nullptr;
    // fp16 overflow word (dir_common.h Ovf): every kernel that packs fp32 sums for a store ORs into it; sticky
    // until dir_engine_overflow() reads and clears it
    int* d_ovf = nullptr;
    // DIR_FP16P on the raw uint8 feed (stem_u8.hip): conv1 + bn1 with ToTensor / Normalize folded in - filter pair, bias and the
    // border-class bias corrections (the desc's mean / std are part of them: a new preprocess means a new engine)
    uint16_t* d_stem_u8_w = nullptr;
    uint16_t* d_stem_u8_w_lo = nullptr;
    float* d_stem_u8_bias = nullptr;
    float* d_stem_u8_corr = nullptr;
    int fold_stem_u8(dir::ConvLayer& L, const float* w, const float* scale, const float* bias);
    // profiling
    bool profiling = false;
    bool prof_paused = false;
    std::vector<dir::ProfSlot> prof;
    size_t prof_used = 0;
    bool tuning = false;

    int build_graph();
    int finalize(int dtype);
    int plan(int B, int H, int W, dir::Plan* p) const;
    int forward(const void* img, int B, int H, int W, int fmt, float* desc_out, void* feat_out,
                int* fh, int* fw, int* fc, void* ws, size_t ws_bytes, hipStream_t stream);
    // the strict fp32 path (conv_f32.hip): the reference's op sequence, nothing fused
    int forward_f32(const void* img, int B, int H, int W, int fmt, float* desc_out, void* feat_out, int* fh, int* fw,
                    int* fc, char* base, const dir::Plan& p, hipStream_t stream);
    int run_conv_f32(dir::ConvLayer& L, const float* x, const float* res, float* y, int B, int H, int W, int OH, int OW,
                     hipStream_t stream);
    int run_conv(dir::ConvLayer& L, const uint16_t* x, const uint16_t* res, uint16_t* y, int B,
                 int H, int W, int OH, int OW, hipStream_t stream, bool rev_m = false);
    // DIR_FP16P head (conv_pair.hip): one convolution on fp16 pairs; x_lo / res_lo / y_lo may be null
    // x2 / x2_lo set: the two-source form (conv3 + the block's stride-1 downsample as one GEMM over [t2 ; block input])
    int run_conv_pair(dir::ConvLayer& L, const uint16_t* x, const uint16_t* x_lo, const uint16_t* res,
                      const uint16_t* res_lo, uint16_t* y, uint16_t* y_lo, int B, int H, int W, int OH, int OW,
                      hipStream_t stream, const uint16_t* x2 = nullptr, const uint16_t* x2_lo = nullptr);
    // image -> stem -> the first pair_blocks residual blocks, everything a pair of fp16 planes; leaves the last block's
    // output (hi plane only: what layer2 reads) in *cur and reports the map size and the next block index
    int forward_pair_head(const void* img, int B, int H, int W, int fmt, char* base, const dir::Plan& p,
                          hipStream_t stream, uint16_t** cur, int* h, int* w, size_t* next_block);
    // its first two launches: image -> paired space-to-depth image -> paired pooled stem output (hi: bufA, lo: lo_stem)
    int forward_pair_stem(const void* img, int B, int H, int W, int fmt, char* base, const dir::Plan& p, hipStream_t stream);
    // conv3 of one bottleneck + conv1 of the next in one kernel (conv_c3c1.hip); *used = 0 when the shapes
    // do not qualify and nothing was launched
    int run_seam(dir::ConvLayer& c3, dir::ConvLayer& c1, const uint16_t* t2, const uint16_t* res, uint16_t* y,
                 uint16_t* t1, int B, int H, int W, hipStream_t stream, int* used,
                 const uint16_t* block_in = nullptr, const uint16_t* block_in_lo = nullptr);
    // conv3 + the block's 1x1 downsample branch as ONE two-source GEMM (conv_persist.hip DUAL form); dry = only
    // report whether it would run (decided before the downsample would be launched)
    int run_conv_dual(dir::ConvLayer& c3, const dir::ConvLayer& ds, const uint16_t* t2, const uint16_t* xin,
                      uint16_t* y, int B, int Hin, int Win, int OH, int OW, hipStream_t stream, int* used, bool dry);
    // A/B switches: the process-wide dir::env() as it stood at dir_engine_create (tests build a new engine after
    // flipping a variable and calling dir_reload_env); forward() never reads the environment
    dir::Env sw;
    float* splitk_scratch = nullptr;  // fp32 partial sums of split-K convs (inside the workspace)
    int prof_begin(const std::string& name, const std::string& kernel, double flops, double bytes,
                   hipStream_t stream);
    int prof_end(hipStream_t stream);
    int overflow(hipStream_t stream, int* overflowed);
    void release();
};
