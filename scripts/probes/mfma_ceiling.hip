// mfma_ceiling.hip - what the bf16 matrix pipe of THIS box sustains with nothing else in the way: every wave keeps its
// operands in registers and issues independent v_mfma_f32_32x32x16_bf16 back to back (8 accumulators, no LDS, no
// memory traffic inside the loop).  Run with random and with all-zero operands and with 1 / 2 / 4 waves per SIMD:
// the gap between the two is the power-dependent part, the numbers are the ceiling any real kernel sits under.
//   hipcc --offload-arch=gfx950 -O3 mfma_ceiling.hip -o mfma_ceiling && ./mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

template <int NT, int NI>   // NI x 2 independent accumulators (NT = 1024 leaves 128 registers per wave: NI = 2)
__global__ void __launch_bounds__(NT) burn(const u32x4_t* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    bf16x8_t a[NI], b[2];
    for (int i = 0; i < NI; ++i) a[i] = __builtin_bit_cast(bf16x8_t, src[(tid * 6 + i) & 65535]);
    for (int i = 0; i < 2; ++i) b[i] = __builtin_bit_cast(bf16x8_t, src[(tid * 6 + 4 + i) & 65535]);
    f32x16_t acc[NI][2];
    for (int i = 0; i < NI; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NI; ++r)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NI; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[tid] = s;
}

static void launch(int wpb, int cus, const u32x4_t* d, float* out, int iters) {
    if (wpb == 256)
        hipLaunchKernelGGL((burn<256, 4>), dim3(cus), dim3(256), 0, 0, d, out, iters);
    else if (wpb == 512)
        hipLaunchKernelGGL((burn<512, 4>), dim3(cus), dim3(512), 0, 0, d, out, iters);
    else
        hipLaunchKernelGGL((burn<1024, 2>), dim3(cus), dim3(1024), 0, 0, d, out, iters);
}

int main() {
    const int N = 65536;
    u32x4_t* h = (u32x4_t*)malloc(N * sizeof(u32x4_t));
    u32x4_t* d;
    float* out;
    hipMalloc(&d, N * sizeof(u32x4_t));
    hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < 4; ++k) {
                // bf16 pairs of magnitude ~1 with random mantissas and signs (mode 0) or zeros (mode 1)
                const uint32_t lo = 0x3f00u | (rand() & 0x80ffu), hi = 0x3f00u | (rand() & 0x80ffu);
                h[i][k] = mode == 0 ? (lo | (hi << 16)) : 0u;
            }
        hipMemcpy(d, h, N * sizeof(u32x4_t), hipMemcpyHostToDevice);
        for (int wpb = 256; wpb <= 1024; wpb *= 2) {   // 4 / 8 / 16 waves per CU = 1 / 2 / 4 per SIMD
            const int iters = 4000;
            launch(wpb, cus, d, out, 200);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            launch(wpb, cus, d, out, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)cus * (wpb / 64) * iters * 32.0 * 2.0 * 32 * 32 * 16;
            printf("%-7s operands, %2d waves/CU: %.2f ms  %.0f TFLOP/s  (%.3f of 2.5 PF)\n", mode == 0 ? "random" : "zero",
                   wpb / 64, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0);
        }
    }
    return 0;
}
