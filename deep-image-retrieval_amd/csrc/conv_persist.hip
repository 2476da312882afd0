// conv_persist.hip — persistent 256x256 workgroups for the wide 1x1 convolutions (gfx950).
//
// The layer3/layer4 1x1 convs (K = 256 ... 2048 -> 256 ... 2048 channels,
// dirtorch/nets/backbones/resnet.py:56,61,70,78,136-141) run best on the 256x256 tile of
// conv_igemm.hip, but that tile allows ONE workgroup per CU (135 KB LDS, ~236 VGPRs), so its three
// phases - first stage fill (~3 us of exposed latency), K loop, epilogue (LDS staging + 128 KB of
// stores) - are strictly serial on a CU.  Here one workgroup per CU walks a strided list of tiles
// and, after the last MFMA of tile i, issues the first K-stage of tile i+1 into ring slot 0 BEFORE
// running tile i's epilogue; the epilogue stages through LDS above slot 0 only.  Fill latency and
// store traffic of consecutive tiles overlap; everything else (descriptor LDS-DMA with scalar K
// offset, XOR-swizzled 128-byte rows, swapped MFMA roles, bias-initialised accumulators, residual
// tile fetched before the K loop) is the conv_igemm design.
//
// LDS map: slot 0 = [0, 64K), slot 1 = [64K, 128K); epilogue staging = [64K, 64K + 68K): 8 waves x
// 32 pixel rows x (64 fp32 + pad), one 64-channel half of a wave's 128 channels at a time.
#include "dir_common.h"
#include "conv_igemm.h"

namespace dir {

static constexpr uint32_t kOOBq = 0x80000000u;

__device__ __forceinline__ void dma16q(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ uint32_t fast_div_q(uint32_t n, uint32_t mul, uint32_t shr) {
    return mul ? (__umulhi(n, mul) >> shr) : n;
}

template <class DT>
__global__ void __launch_bounds__(512) conv1x1_persist_kernel(const ConvArgs a) {
    constexpr int BM = 256, BN = 256, NT = 512;
    constexpr int TM = 2, TN = 4;              // wave tile 64 pixels x 128 channels (4 x 2 waves)
    constexpr int NA = 4, NB = 4;              // DMA instructions per lane per stage (X, W)
    constexpr int XS = BM * 128, STAGE = (BM + BN) * 128;   // 32 KiB + 32 KiB
    constexpr int EROW = 2 * 128 + 16;         // staging row: 64 fp32 + pad
    constexpr int EPI_OFF = STAGE;             // staging lives above slot 0
    typedef typename DT::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int lrow = lane & 31, lhi = lane >> 5;
    const int srcchunk = (tid & 7) ^ ((tid >> 4) & 7);

    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

    const int ntiles = a.tiles_m * a.tiles_n;
    const int T = a.T;

    uint32_t xvoff[NA], wvoff[NB];
    // per-lane DMA offsets of a tile (1x1: one tap, padding-free; the row mask is folded in)
    auto tile_offsets = [&](int tile) {
        const int tile_n = tile % a.tiles_n, tile_m = tile / a.tiles_n;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = tile_m * BM + i * 64 + (tid >> 3);
            uint32_t off;
            if (a.flat) {
                off = (uint32_t)((m * a.Cin + srcchunk * 8) * 2);
            } else {  // strided 1x1 (downsample): output pixel -> input pixel
                const uint32_t mm = m < a.M ? (uint32_t)m : 0u;
                const uint32_t b = fast_div_q(mm, a.div_ohw_mul, a.div_ohw_shr);
                const uint32_t rem = mm - b * (uint32_t)(a.OH * a.OW);
                const uint32_t oh = fast_div_q(rem, a.div_ow_mul, a.div_ow_shr);
                const uint32_t ow = rem - oh * (uint32_t)a.OW;
                off = (uint32_t)((((b * a.H + oh * a.stride) * a.W + ow * a.stride) * a.Cin +
                                  srcchunk * 8) * 2);
            }
            xvoff[i] = m < a.M ? off : kOOBq;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            wvoff[i] = (uint32_t)(((tile_n * BN + i * 64 + (tid >> 3)) * a.Ktot + srcchunk * 8) * 2);
    };
    auto issue = [&](int t, char* stage) {
#pragma unroll
        for (int i = 0; i < NA; ++i) dma16q(rsrc_x, stage + (i * NT + wave * 64) * 16, xvoff[i], t * 128);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            dma16q(rsrc_w, stage + XS + (i * NT + wave * 64) * 16, wvoff[i], t * 128);
    };

    const int lswz = (lane >> 1) & 7;
    int loff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) loff[ks] = lrow * 128 + (((2 * ks + lhi) ^ lswz) << 4);
    const int xfrag = (wm * TM * 32) * 128;
    const int wfrag = XS + (wn * TN * 32) * 128;

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    tile_offsets(tile);
    issue(0, smem);  // first tile: stage 0 -> slot 0

    for (;;) {
        const int tile_n = tile % a.tiles_n, tile_m = tile / a.tiles_n;
        const int m_epi = tile_m * BM + wm * TM * 32;
        const int n_wave = tile_n * BN + wn * TN * 32;

        f32x16_t acc[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t b4 = *(const DIR_GLOBAL f32x4_t*)(a.bias + n_wave + i * 32 + 8 * g + 4 * lhi);
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = b4[e];
            }

        // residual tile of this wave: 2 strips x 2 channel halves x 4 passes of 16 B per lane
        const int ecol = (lane & 7) * 8, erow = lane >> 3;
        u32x4_t rres[TM][2][4];
        if (a.res) {
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int pass = 0; pass < 4; ++pass) {
                        const int m = m_epi + j * 32 + pass * 8 + erow;
                        const int mc = m < a.M ? m : 0;
                        rres[j][h][pass] = gload16(a.res + ((size_t)mc * a.Cout + n_wave + h * 64 + ecol));
                    }
        }

        // ---- K loop: stage 0 is already in flight (issued by the previous tile or the preamble) --
        int issued = 1;
        for (int t = 0; t < T; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // stage t landed everywhere; slot (t+1)&1 is free
            if (issued < T) {
                issue(issued, smem + (issued & 1) * STAGE);
                ++issued;
            }
            const char* stage = smem + (t & 1) * STAGE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                frag_t wf[TN], xf[TM];
#pragma unroll
                for (int i = 0; i < TN; ++i) wf[i] = *(const frag_t*)(stage + wfrag + i * 4096 + loff[ks]);
#pragma unroll
                for (int j = 0; j < TM; ++j) xf[j] = *(const frag_t*)(stage + xfrag + j * 4096 + loff[ks]);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc[i][j] = DT::mfma32(wf[i], xf[j], acc[i][j]);
            }
        }
        __syncthreads();  // both slots are dead

        // ---- next tile's first stage goes out before this tile's epilogue -------------------------
        const int next = tile + (int)gridDim.x;
        if (next < ntiles) {
            tile_offsets(next);
            issue(0, smem);
        }

        // ---- epilogue through the staging area above slot 0 ---------------------------------------
        char* ebase = smem + EPI_OFF + wave * (32 * EROW);
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int i = 2 * h + ii;
                        f32x4_t v = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2],
                                     acc[i][j][4 * g + 3]};
                        *(f32x4_t*)(ebase + lrow * EROW + (ii * 32 + 8 * g + 4 * lhi) * 4) = v;
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int mrow = pass * 8 + erow;
                    const f32x4_t f0 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4);
                    const f32x4_t f1 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4 + 16);
                    const int m = m_epi + j * 32 + mrow;
                    if (m < a.M) {
                        float v[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
                        if (a.res) {
                            const u32x4_t rv = rres[j][h][pass];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float lo, hi;
                                DT::unpack(rv[e], lo, hi);
                                v[2 * e] += lo;
                                v[2 * e + 1] += hi;
                            }
                        }
                        if (a.relu) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        u32x4_t ov;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] = DT::pack(v[2 * e], v[2 * e + 1]);
                        gstore16(a.y + ((size_t)m * a.Cout + n_wave + h * 64 + ecol), ov);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        if (next >= ntiles) break;
        tile = next;
    }
}

bool conv1x1_persist_admissible(const ConvArgs& a) {
    return a.R == 1 && a.S == 1 && a.pad == 0 && a.Cout % 256 == 0 && a.Cin % 64 == 0 && a.Cin >= 128;
}

template <class DT>
static hipError_t launch_persist(const ConvArgs& a, hipStream_t stream) {
    constexpr int LDS = (256 + 256) * 128 + 8 * 32 * (2 * 128 + 16);  // slot 0 + staging (covers slot 1)
    static_assert(LDS >= 2 * (256 + 256) * 128 && LDS <= 160 * 1024, "LDS map");
    auto kern = conv1x1_persist_kernel<DT>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.T = a.Ktot / 64;
    b.tiles_m = ceil_div(a.M, 256);
    b.tiles_n = a.Cout / 256;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    b.flat = (a.stride == 1 && a.H == a.OH && a.W == a.OW);
    // exact n / d for n < 2^31 (same constants as conv_igemm.hip)
    auto fd = [](uint32_t d, uint32_t& mul, uint32_t& shr) {
        if (d <= 1) { mul = 0; shr = 0; return; }
        uint32_t l = 0;
        while ((1ull << l) < d) ++l;
        mul = (uint32_t)(((1ull << (31 + l)) + d - 1) / d);
        shr = l - 1;
    };
    fd((uint32_t)(a.OH * a.OW), b.div_ohw_mul, b.div_ohw_shr);
    fd((uint32_t)a.OW, b.div_ow_mul, b.div_ow_shr);
    const int ntiles = b.tiles_m * b.tiles_n;
    int ncu = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            ncu = prop.multiProcessorCount;
    }
    const int grid = ntiles < ncu ? ntiles : ncu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, b);
    return hipGetLastError();
}

hipError_t conv1x1_persist_launch(const ConvArgs& a, int dtype, hipStream_t stream) {
    return dtype == DIR_BF16 ? launch_persist<BF16>(a, stream) : launch_persist<FP16>(a, stream);
}

}  // namespace dir
