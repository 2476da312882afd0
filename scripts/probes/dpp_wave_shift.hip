// dpp_wave_shift.hip - does gfx950 execute the GFX9 whole-wave DPP shifts (wave_shr:1 = 0x138, wave_shl:1 = 0x130)
// and what do the row-local ones (row_shr:1 = 0x111, row_shl:1 = 0x101) do at the 16-lane row boundaries?
// The u8 stem (csrc/stem_u8.hip) max-pools horizontally in registers with them: lane = conv column.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/dpp_wave_shift.hip -o /tmp/dpp && /tmp/dpp
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int CTRL>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    int v = 100 + lane;
    // old = -1: lanes whose source is outside the wave / row keep old (bound_ctrl = false)
    int r = __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xf, 0xf, false);
    out[lane] = r;
}

template <int CTRL>
static void run(const char* name) {
    int* d;
    int h[64];
    hipMalloc(&d, 256);
    hipLaunchKernelGGL(k<CTRL>, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%s:", name);
    for (int i = 0; i < 64; ++i) printf(" %d", h[i]);
    printf("\n");
    hipFree(d);
}

int main() {
    run<0x138>("wave_shr1");
    run<0x130>("wave_shl1");
    run<0x111>("row_shr1");
    run<0x101>("row_shl1");
    return 0;
}
