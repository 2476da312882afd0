// mall_probe.hip - does a working set that fits the 256 MiB Infinity Cache stream faster than one that lives in HBM?
// (the price of running layers 3-4 depth-first on sub-batches whose tensors stay on die.)
//   rd    : every CU streams the SAME S bytes `passes` times inside one launch (global_load_dwordx4, 8 in flight per lane)
//   chain : ping-pong copies A -> B, B -> A, ... as SEPARATE launches (S/2 read + S/2 written per launch, working set S):
//           the producer -> consumer pattern of consecutive conv launches
//   w2r   : one launch writes S bytes, the next reads them; the READ launch alone is timed (does a store allocate on die?)
//   mix   : the layer3 conv3 shape of traffic: read T2 (S/9) + read RES (4S/9), write OUT (4S/9) IN PLACE over RES
//   hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe && ./mall_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
#define GLOBAL __attribute__((address_space(1)))

__global__ void __launch_bounds__(512) rd(const u32x4_t* __restrict__ p, size_t n, int passes, uint32_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4_t s = {0, 0, 0, 0};
    for (int r = 0; r < passes; ++r) {
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 7 * stride < n; i += 8 * stride) {
            u32x4_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load((const GLOBAL u32x4_t*)(p + i + k * stride)) ;
#pragma unroll
            for (int k = 0; k < 8; ++k) s ^= v[k];
        }
    }
    if ((s[0] ^ s[1] ^ s[2] ^ s[3]) == 0x12345u) out[0] = 1;
}
__global__ void __launch_bounds__(512) rd_plain(const u32x4_t* __restrict__ p, size_t n, int passes, uint32_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4_t s = {0, 0, 0, 0};
    for (int r = 0; r < passes; ++r) {
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 7 * stride < n; i += 8 * stride) {
            u32x4_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *(const GLOBAL u32x4_t*)(p + i + k * stride);
#pragma unroll
            for (int k = 0; k < 8; ++k) s ^= v[k];
        }
    }
    if ((s[0] ^ s[1] ^ s[2] ^ s[3]) == 0x12345u) out[0] = 1;
}
__global__ void __launch_bounds__(512) cp(const u32x4_t* __restrict__ p, u32x4_t* __restrict__ q, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        u32x4_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *(const GLOBAL u32x4_t*)(p + i + k * stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) *(GLOBAL u32x4_t*)(q + i + k * stride) = v[k];
    }
}
__global__ void __launch_bounds__(512) wr(u32x4_t* __restrict__ q, size_t n, uint32_t seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32x4_t v = {seed, seed + 1, seed + 2, (uint32_t)i};
    for (; i < n; i += stride) *(GLOBAL u32x4_t*)(q + i) = v;
}
// conv3-like: out[i] = f(res[i], t2[i / 4]) written over res; n = elements of res
__global__ void __launch_bounds__(512) mix(const u32x4_t* __restrict__ t2, u32x4_t* __restrict__ res, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        u32x4_t v[4];
        const u32x4_t a = *(const GLOBAL u32x4_t*)(t2 + i / 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *(const GLOBAL u32x4_t*)(res + i + k * stride);
#pragma unroll
        for (int k = 0; k < 4; ++k) *(GLOBAL u32x4_t*)(res + i + k * stride) = v[k] ^ a;
    }
}

int main() {
    const size_t cap = 4ull << 30;
    u32x4_t *a, *b;
    uint32_t* out;
    hipMalloc(&a, cap); hipMalloc(&b, cap); hipMalloc(&out, 64);
    hipMemset(a, 1, cap); hipMemset(b, 2, cap);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sizes_mb[] = {24, 48, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 4096};
    printf("# working set S; GB/s = bytes moved (reads + writes) / time; best of 3\n");
    for (size_t mb : sizes_mb) {
        const size_t S = mb << 20, n = S / 16;
        // ---- rd (nt loads) and rd_plain --------------------------------------------------------------------------------
        for (int plain = 0; plain < 2; ++plain) {
            const int passes = (int)((8ull << 30) / S) < 2 ? 2 : (int)((8ull << 30) / S);
            float best = 1e9f;
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(e0, 0);
                if (plain) hipLaunchKernelGGL(rd_plain, dim3(cus), dim3(512), 0, 0, a, n, passes, out);
                else hipLaunchKernelGGL(rd, dim3(cus), dim3(512), 0, 0, a, n, passes, out);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("S %5zu MB  %-9s %3d passes in one launch : %8.3f ms  %6.0f GB/s\n", mb, plain ? "rd" : "rd(nt)", passes, best,
                   (double)S * passes / best / 1e6);
        }
        // ---- chain: A -> B -> A ... separate launches, working set S (two halves) -----------------------------------------
        {
            const size_t h = n / 2;
            const int L = 24;
            float best = 1e9f;
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(e0, 0);
                for (int l = 0; l < L; ++l)
                    hipLaunchKernelGGL(cp, dim3(cus), dim3(512), 0, 0, (l & 1) ? a + h : a, (l & 1) ? a : a + h, h);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("S %5zu MB  chain     %3d copy launches S/2 -> S/2: %8.3f ms  %6.0f GB/s  (%.1f us per launch)\n", mb, L, best,
                   (double)S * L / best / 1e6, best / L * 1e3);
        }
        // ---- w2r: write launch, then read launch; read alone timed ---------------------------------------------------------
        {
            float best = 1e9f, bestw = 1e9f;
            for (int it = 0; it < 4; ++it) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(wr, dim3(cus * 2), dim3(512), 0, 0, b, n, (uint32_t)it);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float msw; hipEventElapsedTime(&msw, e0, e1);
                if (msw < bestw) bestw = msw;
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(rd_plain, dim3(cus), dim3(512), 0, 0, b, n, 1, out);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("S %5zu MB  w2r       write launch %8.3f ms %6.0f GB/s | read-after-write launch %8.3f ms %6.0f GB/s\n", mb, bestw,
                   (double)S / bestw / 1e6, best, (double)S / best / 1e6);
        }
        // ---- mix: conv3-like in-place residual update, repeated launches -----------------------------------------------------
        {
            const size_t nres = n * 4 / 5 / 4 * 4;       // RES = 4/5 of S (read + written in place), T2 = 1/5 of S (read)
            const int L = 16;
            float best = 1e9f;
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(e0, 0);
                for (int l = 0; l < L; ++l) hipLaunchKernelGGL(mix, dim3(cus), dim3(512), 0, 0, a + nres, a, nres);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double moved = (double)nres * 16 * 2.25;   // res read + res written + t2 read
            printf("S %5zu MB  mix       %3d in-place launches        : %8.3f ms  %6.0f GB/s  (%.1f us per launch)\n", mb, L, best,
                   moved * L / best / 1e6, best / L * 1e3);
        }
    }
    return 0;
}
