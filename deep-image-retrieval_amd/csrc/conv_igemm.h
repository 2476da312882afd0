// conv_igemm.h — host-side contract of the implicit-GEMM convolution family (conv_igemm.hip).
#pragma once
#include "dir_common.h"

namespace dir {

enum { STG_GLDS = 0, STG_REG = 1 };

struct ConvArgs {
    const uint16_t* x;     // NHWC [B,H,W,Cin]
    const uint16_t* w;     // [Cout][R][S][Cin]
    const float* bias;     // [Cout]  (folded BatchNorm shift)
    const uint16_t* res;   // NHWC [B,OH,OW,Cout] or nullptr
    uint16_t* y;           // NHWC [B,OH,OW,Cout]
    const uint16_t* zero;  // >= 16 bytes of zeros in device memory (padding source)
    int B, H, W, Cin, OH, OW, Cout;
    int R, S, stride, pad, relu;
    int M;     // B*OH*OW
    int Ktot;  // R*S*Cin
    int T;     // Ktot / 64
    int tiles_m, tiles_n;  // filled by the launcher
};

typedef hipError_t (*ConvLaunchFn)(const ConvArgs&, hipStream_t);

struct ConvVariant {
    const char* name;
    int BM, BN, threads, staging;
    ConvLaunchFn launch[2];    // [dtype]
    ConvLaunchFn launch16[2];  // Cin == 16 stem instantiation, or nullptr
};

int conv_variant_count();
const ConvVariant& conv_variant(int i);
bool conv_variant_admissible(int v, const ConvArgs& a);
int conv_pick_variant(const ConvArgs& a);
int conv_launch(const ConvArgs& a, int dtype, int variant, hipStream_t stream);
int conv_launch_naive(const ConvArgs& a, int dtype, hipStream_t stream);

const uint16_t* zero_page();  // lazily allocated per process (one device per process)

}  // namespace dir
