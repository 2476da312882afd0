// probe: where does `buffer_load_dwordx3 ... lds` put lane i's 12 bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define DIR_LDS __attribute__((address_space(3)))
__global__ void probe(const uint32_t* g, uint32_t* out, int size_sel) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* l = (uint32_t*)smem;
    for (int i = threadIdx.x; i < 1024; i += 64) l[i] = 0xdead0000u + i;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 4096 * 4, 0x00020000);
    const uint32_t voff = threadIdx.x * 32;   // lane i reads dwords 8 i, 8 i + 1, 8 i + 2 [, 8 i + 3]
    if (size_sel == 12)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)smem, 12, voff, 0, 0, 0);
    else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)smem, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = l[i];
}
int main() {
    uint32_t *g, *o;
    hipMalloc(&g, 4096 * 4);
    hipMalloc(&o, 1024 * 4);
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int sz : {12, 16}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, g, o, sz);
        std::vector<uint32_t> r(1024);
        hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
        printf("size %d: first 40 LDS dwords:", sz);
        for (int i = 0; i < 40; ++i) printf(" %x", r[i]);
        printf("\n  dwords 180..200:");
        for (int i = 180; i < 200; ++i) printf(" %x", r[i]);
        printf("\n  dwords 250..262:");
        for (int i = 250; i < 262; ++i) printf(" %x", r[i]);
        printf("\n");
    }
    return 0;
}
