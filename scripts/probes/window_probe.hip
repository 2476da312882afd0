// window_probe.hip - do L2-hit loads share the per-CU request window with HBM misses?
//
// The 1024 -> 256 conv1 of layer3 (conv_persist.hip) moves 32 KB of pixels (HBM) + 32 KB of weights (a 512 KB panel,
// always an L2 hit) per K-step and runs at ~27 GB/s per CU for the sum.  This probe streams the same two kinds of
// LDS-DMA traffic from one 512-thread workgroup per CU and varies (a) who issues what and (b) how many stages are in
// flight, to tell "one in-order window per CU, hits queue behind misses" from "hits are cheap when other waves issue them":
//   px    : 32 KB of unique HBM bytes per step, all 8 waves, 4 DMA instructions per lane
//   px+w  : the same + 32 KB of a shared 512 KB buffer per step, every wave issues both kinds (what the kernel does)
//   split : waves 0-3 issue the pixels (8 per lane), waves 4-7 the weights (8 per lane)
//   w     : the shared buffer only
//   px_d/w1: waves 0-3 keep DEPTH pixel stages in flight, waves 4-7 fetch the weights of step s at step s (one stage)
// DEPTH = stages in flight per wave (the wait before step s leaves DEPTH - 1 newer stages outstanding).
//   hipcc --offload-arch=gfx950 -O3 window_probe.hip -o window_probe && ./window_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define LDSP __attribute__((address_space(3)))

template <int MODE, int DEPTH>
__global__ void __launch_bounds__(512) probe(const char* __restrict__ px, size_t px_per_wg, const char* __restrict__ w,
                                             uint32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int steps = (int)(px_per_wg / 32768);
    const __amdgpu_buffer_rsrc_t rp =
        __builtin_amdgcn_make_buffer_rsrc((void*)(px + (size_t)blockIdx.x * px_per_wg), 0, (int)px_per_wg, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 512 * 1024, 0x00020000);
    constexpr int SLOTS = DEPTH;   // the barrier after the read frees stage s's slot before stage s + DEPTH goes out
    constexpr int PER = (MODE == 1 || MODE == 2 || MODE == 4) ? 8 : 4;   // DMA instructions per lane per step
    uint32_t acc = 0;
    auto issue = [&](int s) {
        char* slot = smem + (s % SLOTS) * 32768 * (MODE == 0 || MODE == 3 ? 1 : 2);
        if (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (LDSP void*)(slot + (k * 8 + wave) * 1024), 16,
                                                         (uint32_t)((k * 8 + wave) * 1024 + lane * 16), s * 32768, 0, 0);
        }
        if (MODE == 1 || MODE == 3) {
            char* ws = slot + (MODE == 1 ? 32768 : 0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (LDSP void*)(ws + (k * 8 + wave) * 1024), 16,
                                                         (uint32_t)((k * 8 + wave) * 1024 + lane * 16), (s & 15) * 32768, 0, 0);
        }
        if (MODE == 2) {
            if (wave < 4) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (LDSP void*)(slot + (k * 4 + wave) * 1024), 16,
                                                             (uint32_t)((k * 4 + wave) * 1024 + lane * 16), s * 32768, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (LDSP void*)(slot + 32768 + (k * 4 + wave - 4) * 1024), 16,
                                                             (uint32_t)((k * 4 + wave - 4) * 1024 + lane * 16), (s & 15) * 32768, 0, 0);
            }
        }
    };
    if (MODE == 4) {
        // deep pixels, shallow weights: waves 0-3 keep DEPTH pixel stages in flight, waves 4-7 fetch step s's weights at step s
        char* wbase = smem + DEPTH * 32768;
        auto issue_px = [&](int s) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (LDSP void*)(smem + (s % DEPTH) * 32768 + (k * 4 + wave) * 1024), 16,
                                                         (uint32_t)((k * 4 + wave) * 1024 + lane * 16), s * 32768, 0, 0);
        };
        if (wave < 4) for (int s = 0; s < DEPTH - 1 && s < steps; ++s) issue_px(s);
        for (int s = 0; s < steps; ++s) {
            if (wave < 4) {
                if (s + DEPTH - 1 < steps) issue_px(s + DEPTH - 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"((DEPTH - 1) * 8) : "memory");
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (LDSP void*)(wbase + (k * 4 + wave - 4) * 1024), 16,
                                                             (uint32_t)((k * 4 + wave - 4) * 1024 + lane * 16), (s & 15) * 32768, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            acc ^= *(const uint32_t*)(smem + (s % DEPTH) * 32768 + wave * 1024 + lane * 4) ^ *(const uint32_t*)(wbase + wave * 1024 + lane * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (acc == 0x12345u) out[0] = 1;
        return;
    }
    for (int s = 0; s < DEPTH - 1 && s < steps; ++s) issue(s);
    for (int s = 0; s < steps; ++s) {
        if (s + DEPTH - 1 < steps) issue(s + DEPTH - 1);
        // step s must have landed; DEPTH - 1 newer stages of PER instructions each may stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"((DEPTH - 1) * PER) : "memory");
        __builtin_amdgcn_s_barrier();   // like the kernels: a stage is consumed by every wave
        acc ^= *(const uint32_t*)(smem + (s % SLOTS) * 32768 * (MODE == 0 || MODE == 3 ? 1 : 2) + wave * 1024 + lane * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // slot free for the stage issued next iteration
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345u) out[0] = 1;
}

// rows: the access pattern of sim_split.hip - per step a workgroup fetches 128 bytes from each of 256 rows of 8 KiB pitch
// (one DMA instruction = 8 rows x 128 B), walking along the rows step by step (64 steps per block of rows, each block
// starting at its own rotation), DEPTH stages in flight.  PIECE = bytes per row per step (128: as the kernel; 256 / 512:
// fewer rows per step, same 32 KB per step) - is the 3.3-3.9 TB/s of that kernel the pattern or the kernel?
template <int DEPTH, int PIECE>
__global__ void __launch_bounds__(512) rows_probe(const char* __restrict__ db, int nblocks, uint32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int ROWS = 32768 / PIECE;            // rows per step
    constexpr int LPR = PIECE / 16;                // lanes per row
    constexpr int STEPS = 8192 / PIECE;            // steps per block of rows
    uint32_t acc = 0;
    int s = 0;
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(db + (size_t)blk * ROWS * 8192), 0, ROWS * 8192, 0x00020000);
        const int rot = (blk * 7) % STEPS;
        uint32_t voff[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = ((k * 8 + wave) * 64 + lane) / LPR;
            voff[k] = (uint32_t)(row * 8192 + (lane % LPR) * 16);
        }
        // (the ring is drained at every block boundary: 64 steps per block make that 3 % of the time at depth 2)
        for (int t = 0; t < DEPTH - 1 && t < STEPS; ++t) {
            const int u = (t + rot) % STEPS;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (LDSP void*)(smem + ((s + t) % DEPTH) * 32768 + (k * 8 + wave) * 1024), 16, voff[k], u * PIECE, 0, 0);
        }
        for (int t = 0; t < STEPS; ++t) {
            if (t + DEPTH - 1 < STEPS) {
                const int u = (t + DEPTH - 1 + rot) % STEPS;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (LDSP void*)(smem + ((s + t + DEPTH - 1) % DEPTH) * 32768 + (k * 8 + wave) * 1024), 16, voff[k], u * PIECE, 0, 0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"((DEPTH - 1) * 4) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            acc ^= *(const uint32_t*)(smem + ((s + t) % DEPTH) * 32768 + wave * 1024 + lane * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        s += STEPS;
    }
    if (acc == 0x12345u) out[0] = 1;
}

template <int DEPTH, int PIECE>
static void run_rows(const char* px, size_t bytes, uint32_t* out, int cus) {
    const int lds = DEPTH * 32768;
    hipFuncSetAttribute((const void*)rows_probe<DEPTH, PIECE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nblocks = (int)(bytes / ((32768 / PIECE) * 8192));
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((rows_probe<DEPTH, PIECE>), dim3(cus), dim3(512), lds, 0, px, nblocks, out);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("rows    depth %d, %3d B per row per step (%3d rows of 8 KiB pitch): %.3f ms  %.0f GB/s\n", DEPTH, PIECE, 32768 / PIECE, best,
           (double)nblocks * (32768 / PIECE) * 8192 / best / 1e6);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

// rd+st: a read stream (16 KB of unique HBM bytes per step per workgroup, LDS-DMA, two steps in flight, issued by waves 0-3)
// next to a store stream from ONE other wave (4 KB per step: the output of a 1x1 convolution tile, 1/4 of the bytes read),
// in three address patterns of the same bytes - what do the stores cost the read stream?
//   PAT 0: no stores            PAT 1: 4 KB contiguous per step (4 instructions of 1 KB)
//   PAT 2: full 128-byte lines, 8 rows of 512 B pitch per instruction (what an LDS-staged epilogue emits)
//   PAT 3: 32 bytes in each of 32 rows of 512 B pitch per instruction (16 B per lane straight from MFMA accumulators)
//   PAT 4: the same bytes as ONE burst per 16 steps: 64 KB contiguous, 16 instructions from each of waves 4-7
template <int PAT>
__global__ void __launch_bounds__(512) rdst_probe(const char* __restrict__ px, size_t px_per_wg, char* __restrict__ out, size_t out_per_wg,
                                                  uint32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int steps = (int)(px_per_wg / 16384);
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    if (wave < 4) {
        const __amdgpu_buffer_rsrc_t rp =
            __builtin_amdgcn_make_buffer_rsrc((void*)(px + (size_t)blockIdx.x * px_per_wg), 0, (int)px_per_wg, 0x00020000);
        auto issue = [&](int s) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (LDSP void*)(smem + (s % 3) * 16384 + (k * 4 + wave) * 1024), 16,
                                                         (uint32_t)((k * 4 + wave) * 1024 + lane * 16), s * 16384, 0, 0);
        };
        issue(0);
        if (steps > 1) issue(1);
        for (int s = 0; s < steps; ++s) {
            if (s + 1 < steps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + 2 < steps) issue(s + 2);
        }
        return;
    }
    u4 v = {(uint32_t)lane, 1u, 2u, 3u};
    char* o = out + (size_t)blockIdx.x * out_per_wg;
    for (int s = 0; s < steps; ++s) {
        __builtin_amdgcn_s_barrier();
        if (PAT == 4) {
            if ((s & 15) == 15) {
                char* tile = o + (size_t)(s >> 4) * 65536;
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    *(__attribute__((address_space(1))) u4*)(tile + (size_t)((wave - 4) * 16 + c) * 1024 + lane * 16) = v;
            }
            continue;
        }
        if (wave != 7 || PAT == 0) continue;
        // the output of the workgroup: tiles of 128 rows x 512 B (64 KB), one tile per 16 steps
        char* tile = o + (size_t)(s >> 4) * 65536;
        const int p = s & 15;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            size_t off;
            if (PAT == 1) off = (size_t)(p * 4 + c) * 1024 + lane * 16;
            else if (PAT == 2) off = (size_t)((p * 4 + c) * 8 + (lane >> 3)) * 512 + (lane & 7) * 16;   // 8 rows x 128 B... of the 512-B row: quarter (p & 3)
            else off = (size_t)((c >> 1) * 64 + (p >> 3) * 32 + (lane & 31)) * 512 + (c & 1) * 256 + ((p >> 1) & 3) * 64 + (p & 1) * 32 + (lane >> 5) * 16;
            if (PAT == 2) off = (size_t)(((p >> 2) * 4 + c) * 8 + (lane >> 3)) * 512 + (p & 3) * 128 + (lane & 7) * 16;
            *(__attribute__((address_space(1))) u4*)(tile + off) = v;
        }
    }
    if (v[0] == 0x12345u) flag[0] = 1;
}

template <int PAT>
static void run_rdst(const char* px, size_t bytes, char* out, uint32_t* flag, int cus) {
    hipFuncSetAttribute((const void*)rdst_probe<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t per_wg = bytes / cus / 262144 * 262144, out_per_wg = per_wg / 4;
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((rdst_probe<PAT>), dim3(cus), dim3(512), 49152, 0, px, per_wg, out, out_per_wg, flag);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const char* names[] = {"no stores", "contiguous 4 KB per step", "128-B lines, 8 rows per instruction", "32 B in each of 32 rows per instruction",
                           "64 KB burst every 16 steps"};
    printf("rd+st   %-40s: %.3f ms  read %.0f GB/s + write %.0f GB/s\n", names[PAT], best, (double)per_wg * cus / best / 1e6,
           PAT ? (double)out_per_wg * cus / best / 1e6 : 0.0);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* px, size_t bytes, const char* w, uint32_t* out, int cus) {
    const int lds = MODE == 4 ? (DEPTH + 1) * 32768 : DEPTH * 32768 * (MODE == 0 || MODE == 3 ? 1 : 2);
    if (lds > 160 * 1024) return;
    hipFuncSetAttribute((const void*)probe<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t per_wg = bytes / cus / 32768 * 32768;
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(cus), dim3(512), lds, 0, px, per_wg, w, out);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double pxb = MODE == 3 ? 0.0 : (double)per_wg * cus, wb = MODE == 0 ? 0.0 : (double)per_wg * cus;
    printf("%-7s depth %d: %.3f ms  HBM pixels %.0f GB/s (%.1f GB/s per CU)  L2 weights %.0f GB/s  sum per CU %.1f GB/s\n", name, DEPTH,
           best, pxb / best / 1e6, pxb / best / 1e6 / cus, wb / best / 1e6, (pxb + wb) / best / 1e6 / cus);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    const size_t bytes = 4ull << 30;
    char *a, *w; uint32_t* out;
    hipMalloc(&a, bytes); hipMalloc(&w, 512 * 1024); hipMalloc(&out, 64);
    hipMemset(a, 1, bytes); hipMemset(w, 2, 512 * 1024);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    char* outb; hipMalloc(&outb, bytes / 4 + (1 << 20));
    for (int rep = 0; rep < 2; ++rep) {
        run_rdst<0>(a, bytes, outb, out, cus); run_rdst<1>(a, bytes, outb, out, cus); run_rdst<2>(a, bytes, outb, out, cus); run_rdst<3>(a, bytes, outb, out, cus); run_rdst<4>(a, bytes, outb, out, cus);
        if (getenv("PROBE_RDST_ONLY")) continue;
        run<0, 1>("px", a, bytes, w, out, cus); run<0, 2>("px", a, bytes, w, out, cus);
        run<0, 3>("px", a, bytes, w, out, cus); run<0, 4>("px", a, bytes, w, out, cus);
        run<1, 1>("px+w", a, bytes, w, out, cus); run<1, 2>("px+w", a, bytes, w, out, cus);
        run<2, 1>("split", a, bytes, w, out, cus); run<2, 2>("split", a, bytes, w, out, cus);
        run<4, 1>("px_d/w1", a, bytes, w, out, cus); run<4, 2>("px_d/w1", a, bytes, w, out, cus);
        run<4, 3>("px_d/w1", a, bytes, w, out, cus); run<4, 4>("px_d/w1", a, bytes, w, out, cus);
        run_rows<2, 128>(a, bytes, out, cus); run_rows<3, 128>(a, bytes, out, cus); run_rows<4, 128>(a, bytes, out, cus);
        run_rows<2, 256>(a, bytes, out, cus); run_rows<2, 512>(a, bytes, out, cus); run_rows<2, 1024>(a, bytes, out, cus);
        run<3, 1>("w", a, bytes, w, out, cus); run<3, 2>("w", a, bytes, w, out, cus); run<3, 4>("w", a, bytes, w, out, cus);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return 1; }
    return 0;
}
