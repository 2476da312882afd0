// conv_f32.h — host-side contract of the strict fp32 path (conv_f32.hip).
#pragma once
#include "dir_common.h"

namespace dir {

struct ConvF32Args {
    const float* x;     // NHWC [B,H,W,Cin]
    const float* w;     // [Cout][R][S][Cin], eval-mode BatchNorm folded in
    const float* bias;  // [Cout]
    const float* res;   // NHWC [B,OH,OW,Cout] or nullptr
    float* y;           // NHWC [B,OH,OW,Cout]
    int B, H, W, Cin, OH, OW, Cout;
    int R, S, stride, pad, relu;
    int M;              // B*OH*OW
    int Ktot;           // R*S*Cin
    int tiles_n;        // filled by the launcher
};

// variant (optional): receives the tile name ("128x128" / "128x64") for the profile label
int conv_f32_launch(ConvF32Args a, hipStream_t stream, const char** variant = nullptr);
int prep_input_f32(const void* img, int fmt, const float* mean3, const float* std3, float* out, int B, int H, int W,
                   hipStream_t stream);
int maxpool_3x3s2_f32(const float* x, float* y, int B, int H, int W, int C, hipStream_t stream);
int global_pool_f32(const float* x, float* out, int ldo, int B, int H, int W, int C, int pooling, float p, float eps,
                    float center_bias, hipStream_t stream);
int upsample_add_f32(const float* x, const float* low, float* y, int B, int H, int W, int h, int w, int C,
                     hipStream_t stream);

}  // namespace dir
