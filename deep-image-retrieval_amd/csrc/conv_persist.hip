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

// XDEEP = false: the map above (2 slots of X+W).
// XDEEP = true : THREE slots for the pixel operand, two for the weights, K-step still 64:
//     X slots [0,32K) [32K,64K) [64K,96K), W slots [96K,128K) [128K,160K); staging [32K,100K).
//   The weight half of every stage is an L2 hit, the pixel half comes from HBM: with one 64 KB stage in
//   flight only 32 KB per CU are HBM requests (round 2 read the 3.9 TB/s of layer3's 1024 -> 256 convs as that
//   window's limit; round 3's probes put the limit elsewhere - waves that both issue LDS-DMA and multiply
//   serialise the two, and the layer's read/write mix caps HBM at ~5.2 TB/s: conv_ring.hip).  Here
//   X runs two K-steps ahead and W one (issue order W(t+1), X(t+2); the wait before step t leaves exactly
//   X(t+1) - the newest ops, all of the LDS-DMA kind - in flight), so 64-96 KB of HBM requests per CU
//   are outstanding.  Next tile's W(0) / X(0) still go out before the epilogue, X(1) right after it.
// DUAL (deep-X form only, round 4): the K dimension comes from TWO tensors - K-steps [0, Cin/64) from x (flat: the
// conv3 input t2), the rest from x2 (the block input, gathered at stride2: the 1x1 downsample) - against weights and
// biases concatenated / summed at finalize: relu([W3 | Wds] . [t2 ; x_s] + b3 + bds), the first block of layers 2-4
// (resnet.py:78-85 with :134-141) as ONE persistent GEMM.  Rounds 2-3 ran the same GEMM on conv_igemm.hip's 256x256 tile,
// one tile per workgroup: fill and epilogue exposed on every one of its short (6-24 K-step) tiles, 5-13 % slower.
template <class DT, bool XDEEP, bool RES, bool DUAL = false>
__global__ void __launch_bounds__(512) conv1x1_persist_kernel(const ConvArgs a) {
    static_assert(!DUAL || (XDEEP && !RES), "the two-source form rides on the deep-X ring");
    constexpr int BM = 256, BN = 256, NT = 512;
    constexpr int TM = 2, TN = 4;              // wave tile 64 pixels x 128 channels (4 x 2 waves)
    constexpr int NA = 4, NB = 4;              // DMA instructions per lane per stage (X, W)
    constexpr int XS = BM * 128, STAGE = (BM + BN) * 128;   // 32 KiB + 32 KiB
    constexpr int EROW = 2 * 128 + 16;         // staging row: 64 fp32 + pad
    constexpr int EPI_OFF = XDEEP ? XS : STAGE;   // staging: above slot 0 / above X slot 0
    constexpr int WOFF = 3 * XS;               // XDEEP: first weight slot
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
    const __amdgpu_buffer_rsrc_t rsrc_x2 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? a.x2 : a.x), 0, DUAL ? a.x2_bytes : a.x_bytes, 0x00020000);

    const int ntiles = a.tiles_m * a.tiles_n;
    const int T = a.T;
    const int T1 = DUAL ? a.Cin / 64 : 0;      // K-steps served by the first source

    uint32_t xvoff[NA], wvoff[NB], xvoff2[DUAL ? NA : 1];
    // per-lane DMA offsets of a tile (1x1: one tap, padding-free; the row mask is folded in)
    auto tile_offsets = [&](int tile) {
        const int tile_n = tile % a.tiles_n, tile_m = a.rev_m ? a.tiles_m - 1 - tile / a.tiles_n : tile / a.tiles_n;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = tile_m * BM + i * 64 + (tid >> 3);
            uint32_t off;
            if (a.flat) {
#ifdef DIR_PERSIST_BLOCKED   // experiment builds only (timing): X read as [M / 256][Cin / 64][256 px][64 ch] - every K-stage one contiguous 32 KB
                off = (uint32_t)(tile_m * (BM * a.Cin * 2) + (i * 64 + (tid >> 3)) * 128 + srcchunk * 16);
#else
                off = (uint32_t)((m * a.Cin + srcchunk * 8) * 2);
#endif
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
            if (DUAL) {   // output pixel -> pixel (oh * stride2, ow * stride2) of the second source
                const uint32_t mm = m < a.M ? (uint32_t)m : 0u;
                const uint32_t b = fast_div_q(mm, a.div_ohw_mul, a.div_ohw_shr);
                const uint32_t rem = mm - b * (uint32_t)(a.OH * a.OW);
                const uint32_t oh = fast_div_q(rem, a.div_ow_mul, a.div_ow_shr);
                const uint32_t ow = rem - oh * (uint32_t)a.OW;
                const uint32_t off2 = (uint32_t)((((b * a.H2 + oh * a.stride2) * a.W2 + ow * a.stride2) * a.Cin2 +
                                                  srcchunk * 8) * 2);
                xvoff2[DUAL ? i : 0] = m < a.M ? off2 : kOOBq;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            wvoff[i] = (uint32_t)(((tile_n * BN + i * 64 + (tid >> 3)) * a.Ktot + srcchunk * 8) * 2);
    };
    auto issue_x = [&](int t, char* dst) {
        if (DUAL && t >= T1) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                dma16q(rsrc_x2, dst + (i * NT + wave * 64) * 16, xvoff2[DUAL ? i : 0], (t - T1) * 128);
            return;
        }
#pragma unroll
#ifdef DIR_PERSIST_BLOCKED
        for (int i = 0; i < NA; ++i) dma16q(rsrc_x, dst + (i * NT + wave * 64) * 16, xvoff[i], a.flat ? t * (BM * 128) : t * 128);
#else
        for (int i = 0; i < NA; ++i) dma16q(rsrc_x, dst + (i * NT + wave * 64) * 16, xvoff[i], t * 128);
#endif
    };
    auto issue_w = [&](int t, char* dst) {
#pragma unroll
        for (int i = 0; i < NB; ++i) dma16q(rsrc_w, dst + (i * NT + wave * 64) * 16, wvoff[i], t * 128);
    };
    auto issue = [&](int t, char* stage) {
        issue_x(t, stage);
        issue_w(t, stage + XS);
    };
    // XDEEP slot addresses: X(t) -> slot t % 3, W(t) -> slot (t + 1) & 1 (so a tile's W(0) sits in the
    // slot the staging area never touches)
    auto xslot = [&](int t) { return smem + (t % 3) * XS; };
    auto wslot = [&](int t) { return smem + WOFF + ((t + 1) & 1) * XS; };

    const int lswz = (lane >> 1) & 7;
    int loff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) loff[ks] = lrow * 128 + (((2 * ks + lhi) ^ lswz) << 4);
    const int xfrag = (wm * TM * 32) * 128;
    const int wfrag = (XDEEP ? 0 : XS) + (wn * TN * 32) * 128;

    // Hardware places workgroup b on XCD b % 8.  With tile = b the tiles_n channel tiles of one pixel tile would sit
    // on tiles_n DIFFERENT XCDs, each fetching the same pixels from HBM into its own L2 (layer4's 512 -> 2048 conv3:
    // PMC traffic 1.8x the algorithmic bytes).  The bijective remap gives every XCD a contiguous run of tiles, so the
    // channel tiles of a pixel tile share one L2; tile + gridDim.x keeps that property round after round.
    int tile = a.no_xcd_map ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= ntiles) return;
    Ovf<DT> ovf;
    // The bias of a tile (its accumulators start there) is fetched BEFORE that tile's first DMA goes
    // out: the compiler waits for ordinary loads in program order, so a bias load issued after the
    // prefetches would drag a wait for all of them to the top of every tile.
    // (64 registers: only the residual-free deep-X form has them to spare; the other form loads the bias
    // at the top of the tile as before.)
    f32x4_t bz[XDEEP ? TN : 1][XDEEP ? 4 : 1];
    auto load_bias = [&](int t) {
        if (!XDEEP) return;
        const int nw = (t % a.tiles_n) * BN + wn * TN * 32;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bz[XDEEP ? i : 0][XDEEP ? g : 0] = *(const DIR_GLOBAL f32x4_t*)(a.bias + nw + i * 32 + 8 * g + 4 * lhi);
    };
    load_bias(tile);
    tile_offsets(tile);
    if (XDEEP) {
        issue_w(0, wslot(0));
        issue_x(0, xslot(0));
        if (T > 1) issue_x(1, xslot(1));
    } else {
        issue(0, smem);  // first tile: stage 0 -> slot 0
    }

    for (;;) {
        const int tile_n = tile % a.tiles_n, tile_m = a.rev_m ? a.tiles_m - 1 - tile / a.tiles_n : tile / a.tiles_n;
        const int m_epi = tile_m * BM + wm * TM * 32;
        const int n_wave = tile_n * BN + wn * TN * 32;

        f32x16_t acc[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t b4 = XDEEP ? bz[XDEEP ? i : 0][XDEEP ? g : 0]
                                         : *(const DIR_GLOBAL f32x4_t*)(a.bias + n_wave + i * 32 + 8 * g + 4 * lhi);
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = b4[e];
            }

        // residual tile of this wave: 2 strips x 2 channel halves x 4 passes of 16 B per lane
        const int ecol = (lane & 7) * 8, erow = lane >> 3;
        // RES = false instantiations (the deep-X form) carry no residual registers: 64 VGPRs less
        u32x4_t rres[RES ? TM : 1][2][RES ? 4 : 1];
        if (RES && a.res) {
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int pass = 0; pass < 4; ++pass) {
                        const int m = m_epi + j * 32 + pass * 8 + erow;
                        const int mc = m < a.M ? m : 0;
                        rres[RES ? j : 0][h][RES ? pass : 0] = gload16(a.res + ((size_t)mc * a.Cout + n_wave + h * 64 + ecol));
                    }
        }

        // ---- K loop: stage 0 is already in flight (issued by the previous tile or the preamble) --
        if (XDEEP) {
            for (int t = 0; t < T; ++t) {
                // need X(t), W(t); may leave X(t+1) - the 4 newest ops, same kind - in flight.  With a
                // residual the tile's first wait also covers the (newer, VGPR-kind) residual loads.
                if (t + 1 < T && !(RES && t == 0 && a.res)) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NA) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                ring_barrier();  // step t landed everywhere; X slot (t+2)%3 and W slot of t+1 are free
                if (t + 1 < T) issue_w(t + 1, wslot(t + 1));
                if (t + 2 < T) issue_x(t + 2, xslot(t + 2));
                const char* xs = xslot(t);
                const char* ws = wslot(t);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    frag_t wf[TN], xf[TM];
#pragma unroll
                    for (int i = 0; i < TN; ++i) wf[i] = *(const frag_t*)(ws + wfrag + i * 4096 + loff[ks]);
#pragma unroll
                    for (int j = 0; j < TM; ++j) xf[j] = *(const frag_t*)(xs + xfrag + j * 4096 + loff[ks]);
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TM; ++j) acc[i][j] = DT::mfma32(wf[i], xf[j], acc[i][j]);
                }
            }
        } else {
        int issued = 1;
        for (int t = 0; t < T; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ring_barrier();  // stage t landed everywhere; slot (t+1)&1 is free
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
        }
        __syncthreads();  // every slot is dead

        // ---- next tile's first stage goes out before this tile's epilogue -------------------------
        const int next = tile + (int)gridDim.x;
        if (next < ntiles) {
            load_bias(next);
            tile_offsets(next);
            if (XDEEP) {
                issue_w(0, wslot(0));
                issue_x(0, xslot(0));
            } else {
                issue(0, smem);
            }
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
                        if (RES && a.res) {
                            const u32x4_t rv = rres[RES ? j : 0][h][RES ? pass : 0];
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
                        ovf.see(ov);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        if (next >= ntiles) break;
        if (XDEEP && T > 1) {
            // every wave is done with the staging area (it covers X slot 1).  Raw barrier: __syncthreads()
            // would drain vmcnt, i.e. wait for the stage already in flight and for this tile's stores.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ring_barrier();
            issue_x(1, xslot(1));
        }
        tile = next;
    }
    ovf.flush(a.ovf);
}

bool conv1x1_persist_admissible(const ConvArgs& a) {
    return a.R == 1 && a.S == 1 && a.pad == 0 && a.Cout % 256 == 0 && a.Cin % 64 == 0 && a.Cin >= 128;
}

template <class DT, bool XDEEP, bool DUAL = false>
static hipError_t launch_persist(const ConvArgs& a, hipStream_t stream) {
    // 2-slot map: slot 0 + staging (covers slot 1); deep-X map: 3 X slots + 2 W slots = all 160 KiB
    constexpr int LDS = XDEEP ? 5 * 256 * 128 : (256 + 256) * 128 + 8 * 32 * (2 * 128 + 16);
    static_assert(LDS >= 2 * (256 + 256) * 128 && LDS <= 160 * 1024, "LDS map");
    static_assert(!XDEEP || 256 * 128 + 8 * 32 * (2 * 128 + 16) <= 3 * 256 * 128 + 256 * 128, "staging must end below W slot 1");
    auto kern = conv1x1_persist_kernel<DT, XDEEP, !XDEEP, DUAL>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.T = a.Ktot / 64;
    b.tiles_m = ceil_div(a.M, 256);
    b.tiles_n = a.Cout / 256;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    if (DUAL) b.x2_bytes = (uint32_t)((size_t)a.B * a.H2 * a.W2 * a.Cin2 * 2);
    b.flat = (a.stride == 1 && a.H == a.OH && a.W == a.OW);
    const bool no_xcd_map = env().no_xcdmap;   // A/B and bisecting
    b.no_xcd_map = no_xcd_map;
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
    const int ncu = cu_count();
    const int grid = ntiles < ncu ? ntiles : ncu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, b);
    return hipGetLastError();
}

// the two-source form: a = the conv3 (1x1 stride 1 over t2) with x2 / Cin2 / H2 / W2 / stride2 set (ConvArgs in conv_igemm.h)
hipError_t conv1x1_persist_dual_bf16(const ConvArgs& a, hipStream_t stream) { return launch_persist<BF16, true, true>(a, stream); }
hipError_t conv1x1_persist_dual_fp16(const ConvArgs& a, hipStream_t stream) { return launch_persist<FP16, true, true>(a, stream); }

hipError_t conv1x1_persist_launch(const ConvArgs& a, int dtype, hipStream_t stream, bool xdeep) {
    if (xdeep) return dtype == DIR_BF16 ? launch_persist<BF16, true>(a, stream) : launch_persist<FP16, true>(a, stream);
    return dtype == DIR_BF16 ? launch_persist<BF16, false>(a, stream) : launch_persist<FP16, false>(a, stream);
}

}  // namespace dir
