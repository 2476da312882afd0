// Microbenchmark: do a wave's ds_read_b128 wait behind its own outstanding LDS-DMA loads?
//   mode 0: ds_read loop alone            mode 1: 8 LDS-DMA loads (1 KB each) issued first, then the loop
//   mode 2: like 1, but the DMA is issued by the OTHER half of the workgroup's waves
// Prints cycles (s_memtime) for the ds_read loop and for "loop + wait for the DMA".
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds, uint32_t voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (LDS_AS void*)lds, 16, voff, soff, 0, 0);
}

__global__ void __launch_bounds__(512) bench(const char* src, size_t src_bytes, long long* out, int mode,
                                             int nload, int nread) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (uint32_t)src_bytes, 0x00020000);
    // region A (DMA target): 64 KB at offset 64 KB; region B (read by ds_read): first 64 KB
    for (int i = threadIdx.x; i < 16384; i += 512) ((uint32_t*)smem)[i] = i;
    __syncthreads();
    const bool issuer = mode == 1 || (mode == 2 && wave >= 4);
    const uint32_t base = (blockIdx.x * 8 + wave) * 65536u + lane * 16u;
    long long t0 = __builtin_readcyclecounter();
    if (issuer)
        for (int i = 0; i < nload; ++i) dma16(rsrc, smem + 65536 + (wave * nload + i) * 1024, base + i * 1024u, 0);
    long long t1 = __builtin_readcyclecounter();
    u32x4 acc = {0, 0, 0, 0};
    uint32_t off = (wave * 64 + lane) * 16;
    const bool reader = mode != 2 || wave < 4;
    if (reader)
        for (int i = 0; i < nread; ++i) {
            const u32x4 v = *(const u32x4*)(smem + ((off + i * 8192) & 65535));
            acc += v;
            off += v[0] & 16;   // dependent chain: every read waits for the previous one
        }
    long long t2 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t3 = __builtin_readcyclecounter();
    if (lane == 0) {
        long long* o = out + ((size_t)blockIdx.x * 8 + wave) * 4;
        o[0] = t1 - t0;
        o[1] = t2 - t1;
        o[2] = t3 - t1;
        o[3] = acc[0] + acc[1] + acc[2] + acc[3];
    }
}

int main() {
    const int nblk = 256;
    const size_t bytes = (size_t)nblk * 8 * 65536 + 65536;
    char* src;
    long long *out, *h = (long long*)malloc(nblk * 8 * 4 * sizeof(long long));
    hipMalloc(&src, bytes);
    hipMemset(src, 1, bytes);
    hipMalloc(&out, nblk * 8 * 4 * sizeof(long long));
    hipFuncSetAttribute((const void*)bench, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int nread : {16, 64, 256})
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(bench, dim3(nblk), dim3(512), 131072, 0, src, bytes, out, mode, 8, nread);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, out, nblk * 8 * 4 * sizeof(long long), hipMemcpyDeviceToHost);
            double s[3] = {0, 0, 0}, sr[3] = {0, 0, 0};
            int nr = 0, ni = 0;
            for (int w = 0; w < nblk * 8; ++w) {
                const bool reader = mode != 2 || (w % 8) < 4;
                for (int k = 0; k < 3; ++k) (reader ? sr : s)[k] += h[w * 4 + k];
                reader ? ++nr : ++ni;
            }
            printf("nread %3d mode %d | reader waves: issue %.0f  ds_read loop %.0f  loop+vmcnt(0) %.0f", nread, mode,
                   sr[0] / nr, sr[1] / nr, sr[2] / nr);
            if (ni) printf(" | issuer waves: issue %.0f  until vmcnt(0) %.0f", s[0] / ni, s[2] / ni);
            printf("   (cycle-counter units)\n");
        }
    return 0;
}
