// conv_patch.hip — 3x3 stride-1 convolution from an LDS-resident input patch (gfx950).
//
// For the narrow 3x3 layers (layer1: 64->64 at 256x256, layer2: 128->128 at 128x128 for a 1024^2
// image; dirtorch/nets/backbones/resnet.py:58-59,74) the implicit-GEMM kernel re-fetches the
// activation tile once per filter tap: 9x the input through L2 and per-tap address arithmetic.
// Here a workgroup loads the (8+2) x (32+2) pixel input patch of its 8 x 32 output tile ONCE
// (LDS-DMA, bounds-checked descriptor = zero padding) and the nine taps read it at shifted pixel
// offsets; only the weights (8/16 KiB per tap) stream through a 3-slot ring.  L2->LDS traffic drops
// from 9x to 1.33x the input, HBM traffic is unchanged (in + out once).
//
// LDS patch: one 128-byte row per pixel per 64-channel plane, 16-byte chunks XOR-swizzled with
// ((p >> 1) & 7), p = linear patch pixel; a 32-pixel MFMA tile is 32 consecutive p, so the
// ds_read_b128 fragment reads stay bank-conflict free for every tap shift.
// MFMA roles, bias-initialised accumulators and the LDS-staged 16-byte-store epilogue are those of
// conv_igemm.hip; the weight layout [Cout][3][3][Cin] is shared with it.
#include "dir_common.h"
#include "conv_igemm.h"

namespace dir {

static constexpr uint32_t kOOBp = 0x80000000u;

__device__ __forceinline__ void dma16p(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)lds, 16, voff, soff, 0, 0);
}

// NTH = 256: four waves, each two output rows x all of Cout.  NTH = 512 (the 128-channel layers):
// eight waves, the second four take the upper half of Cout - same LDS image, two waves per SIMD.
template <class DT, int CIN, int COUT, int NTH>
__global__ void __launch_bounds__(NTH) conv_patch3x3_kernel(const ConvArgs a) {
    constexpr int TH = 8, TW = 32;           // output tile
    constexpr int PH = TH + 2, PW = TW + 2;  // input patch
    constexpr int PP = PH * PW;              // 340 patch pixels
    constexpr int KC = CIN / 64;             // 64-channel planes
    constexpr int WGN = NTH / 256;           // wave groups along Cout
    constexpr int NPL = (PP * 8 + NTH - 1) / NTH;      // DMA instructions per lane per plane (11 / 6)
    constexpr int PLANE_BYTES = NPL * NTH * 16;        // whole DMA instructions land inside
    constexpr int TN = COUT / 32 / WGN;      // channel tiles per wave
    constexpr int TMR = TH / 4;              // output rows per wave (one 32-pixel MFMA tile each)
    constexpr int NSTW = 3;                  // weight ring depth
    constexpr int WSTAGE = COUT * 128;       // one tap's [Cout][64] slice
    constexpr int NBW = COUT * 8 / NTH;      // weight DMA instructions per lane per stage
    constexpr int T = 9 * KC;
    constexpr int WOFF = KC * PLANE_BYTES;   // ring starts after the patch
    constexpr int EROW = TN * 128 + 16;
    typedef typename DT::frag_t frag_t;
    static_assert(CIN % 64 == 0 && COUT % (32 * WGN) == 0 && (NTH == 256 || NTH == 512), "shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31;
    const int lhi = lane >> 5;
    const int wrow = wave & 3;   // which pair of output rows
    const int wn = wave >> 2;    // which slice of Cout

    // tile coordinates: blockIdx.x -> (b, tile_y, tile_x), x fastest
    const int tiles_x = (a.OW + TW - 1) / TW;
    const int tiles_y = (a.OH + TH - 1) / TH;
    int wg = blockIdx.x;
    const int tx = wg % tiles_x;
    wg /= tiles_x;
    const int ty = wg % tiles_y;
    const int b = wg / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;

    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

    // ---- patch: PP pixels x KC planes, loaded once -----------------------------------------------
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int P = i * NTH + tid;   // chunk slot in the plane image
        const int p = P >> 3;          // patch pixel
        const int slot = P & 7;
        const int py = p / PW, px = p - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool ok = p < PP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const uint32_t v = ok ? (uint32_t)((((b * a.H + iy) * a.W + ix) * CIN +
                                            ((slot ^ ((p >> 1) & 7)) << 3)) * 2)
                              : kOOBp;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
            dma16p(rsrc_x, smem + kc * PLANE_BYTES + (i * NTH + wave * 64) * 16, v, kc * 128);
    }

    // ---- weights: one [Cout][64] slice per K-step through an NSTW-slot ring ----------------------
    const int srcchunk = (tid & 7) ^ ((tid >> 4) & 7);
    uint32_t wvoff[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i)
        wvoff[i] = (uint32_t)(((i * (NTH / 8) + (tid >> 3)) * a.Ktot + srcchunk * 8) * 2);
    auto issue_w = [&](int t, int slot) {
#pragma unroll
        for (int i = 0; i < NBW; ++i)
            dma16p(rsrc_w, smem + WOFF + slot * WSTAGE + (i * NTH + wave * 64) * 16, wvoff[i], t * 128);
    };

    // ---- accumulators start at the bias ------------------------------------------------------------
    f32x16_t acc[TN][TMR];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_t b4 = *(const DIR_GLOBAL f32x4_t*)(a.bias + (wn * TN + i) * 32 + 8 * g + 4 * lhi);
#pragma unroll
            for (int j = 0; j < TMR; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = b4[e];
        }

    const int wswz = (lane >> 1) & 7;
    int woffk[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) woffk[ks] = lrow * 128 + (((2 * ks + lhi) ^ wswz) << 4);

    int issued = 0;
#pragma unroll
    for (int p = 0; p < NSTW - 1; ++p) {
        issue_w(p, p);
        ++issued;
    }
    int slot_c = 0, slot_i = NSTW - 1;
    int r = 0, s = 0, kc = 0;  // tap / plane of the step being COMPUTED
    for (int t = 0; t < T; ++t) {
        const int ahead = issued - 1 - t;
        if (ahead >= 1) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NBW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ring_barrier();  // patch (issued first) and weight stage t have landed
        if (issued < T) {
            issue_w(issued, slot_i);
            ++issued;
            if (++slot_i == NSTW) slot_i = 0;
        }
        const char* wst = smem + WOFF + slot_c * WSTAGE;
        const char* plane = smem + kc * PLANE_BYTES;
        frag_t xf[TMR][4];
#pragma unroll
        for (int j = 0; j < TMR; ++j) {
            const int p = (wrow * TMR + j + r) * PW + s + lrow;  // patch pixel read by this lane
            const int swz = (p >> 1) & 7;
            const char* row = plane + p * 128;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                xf[j][ks] = *(const frag_t*)(row + (((2 * ks + lhi) ^ swz) << 4));
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            frag_t wf[TN];
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = *(const frag_t*)(wst + (wn * TN + i) * 4096 + woffk[ks]);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TMR; ++j) acc[i][j] = DT::mfma32(wf[i], xf[j][ks], acc[i][j]);
        }
        if (++slot_c == NSTW) slot_c = 0;
        if (++kc == KC) {
            kc = 0;
            if (++s == 3) {
                s = 0;
                ++r;
            }
        }
    }
    __syncthreads();  // every wave is done with the patch before it becomes epilogue staging
    Ovf<DT> ovf;

    // ---- epilogue (as conv_igemm): acc -> LDS fp32 -> ReLU -> 16-byte stores ------------------------
    char* ebase = smem + wave * (32 * EROW);
    constexpr int LPR = TN * 4, RPP = 64 / LPR, NPASS = 32 / RPP;
    const int ecol = (lane % LPR) * 8;
    const int erow = lane / LPR;
    const int ncol = wn * TN * 32 + ecol;   // first of this lane's 8 output channels
#pragma unroll
    for (int j = 0; j < TMR; ++j) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4_t v = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2],
                             acc[i][j][4 * g + 3]};
                *(f32x4_t*)(ebase + lrow * EROW + (i * 32 + 8 * g + 4 * lhi) * 4) = v;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int oy = oy0 + wrow * TMR + j;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int mrow = pass * RPP + erow;
            const f32x4_t f0 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4);
            const f32x4_t f1 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4 + 16);
            const int ox = ox0 + mrow;
            if (oy < a.OH && ox < a.OW) {
                float v[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
                const size_t o = ((size_t)(b * a.OH + oy) * a.OW + ox) * COUT + ncol;
                if (a.res) {
                    const u32x4_t rv = gload16(a.res + o);
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
                gstore16(a.y + o, ov);
                ovf.see(ov);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    ovf.flush(a.ovf);
}

bool conv_patch3x3_admissible(const ConvArgs& a) {
    return a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.H == a.OH && a.W == a.OW &&
           a.Cin == a.Cout && (a.Cin == 64 || a.Cin == 128);
}

template <class DT, int C, int NTH>
static hipError_t launch_patch(const ConvArgs& a, hipStream_t stream) {
    constexpr int KC = C / 64;
    constexpr int PLANE_BYTES = ((10 * 34 * 8 + NTH - 1) / NTH) * NTH * 16;
    constexpr int TN = C / 32 / (NTH / 256);
    constexpr int MAIN = KC * PLANE_BYTES + 3 * C * 128;
    constexpr int EPI = (NTH / 64) * 32 * (TN * 128 + 16);
    constexpr int LDS = MAIN > EPI ? MAIN : EPI;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv_patch3x3_kernel<DT, C, C, NTH>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    const long blocks = (long)a.B * ((a.OH + 7) / 8) * ((a.OW + 31) / 32);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NTH), LDS, stream, b);
    return hipGetLastError();
}

hipError_t conv_patch3x3_launch(const ConvArgs& a, int dtype, hipStream_t stream) {
    if (a.Cin == 64)
        return dtype == DIR_BF16 ? launch_patch<BF16, 64, 256>(a, stream) : launch_patch<FP16, 64, 256>(a, stream);
    return dtype == DIR_BF16 ? launch_patch<BF16, 128, 512>(a, stream) : launch_patch<FP16, 128, 512>(a, stream);
}


// ---- wide layers: the patch one 64-channel plane at a time ---------------------------------------------
// layer3 / layer4 (256 -> 256, 512 -> 512 at 64^2 / 32^2 for a 1024^2 image): the whole-depth patch does not
// fit (4 or 8 planes of 48 KB), so the K loop goes plane-major - for each 64-channel plane: its 10 x 34
// patch, then nine weight stages [256][64] through a 3-slot ring.  What bounds the implicit-GEMM form on
// these layers is neither MFMA nor bandwidth but the LATENCY of a stage under load: ~1.8 us from issue to
// landed against 0.86 us of MFMA work per 256 x 256 x 64 K-step, with room for only ONE 64 KB stage in
// flight (profiles/r02: waves parked 48 % of their cycles, MfmaUtil 0.53).  Streaming only the weights
// (32 KB per K-step; the patch is 5.3 KB per K-step amortised) leaves LDS for TWO stages in flight:
// 48 KB plane + 3 x 32 KB = 144 KB.  The plane is single-buffered - the next one is requested when the
// last tap of this one is done, with the first two weight stages of the next plane already in flight.
// Cout is tiled by 256 (blockIdx fastest, neighbours share the patch through L2).
template <class DT, int CIN>
__global__ void __launch_bounds__(512) conv_patch3x3s_kernel(const ConvArgs a) {
    constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2, PP = PH * PW;
    constexpr int KC = CIN / 64, NTH = 512, BN = 256;
    constexpr int NPL = (PP * 8 + NTH - 1) / NTH;       // 6 DMA instructions per lane per plane
    constexpr int PLANE_BYTES = NPL * NTH * 16;         // 49152
    constexpr int TN = 4, TMR = 2;                      // wave tile: 128 channels x 2 output rows
    constexpr int WSTAGE = BN * 128, NBW = BN * 8 / NTH;   // 32 KB, 4 instructions per lane
    constexpr int NSTW = 3;
    constexpr int WOFF = PLANE_BYTES;
    constexpr int T = 9 * KC;
    constexpr int EROW = TN * 128 + 16;
    typedef typename DT::frag_t frag_t;
    static_assert(CIN % 64 == 0 && WOFF + NSTW * WSTAGE <= 160 * 1024, "shape / LDS map");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int wrow = wave & 3, wn = wave >> 2;

    const int tiles_n = a.Cout / BN;
    const int tiles_x = (a.OW + TW - 1) / TW;
    const int tiles_y = (a.OH + TH - 1) / TH;
    int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % tiles_n;
    wg /= tiles_n;
    const int tx = wg % tiles_x;
    wg /= tiles_x;
    const int ty = wg % tiles_y;
    const int b = wg / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

    // patch: per-lane source offsets of plane 0 (plane kc adds kc * 128 bytes through the scalar offset)
    uint32_t pvoff[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int P = i * NTH + tid;
        const int p = P >> 3, slot = P & 7;
        const int py = p / PW, px = p - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool ok = p < PP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        pvoff[i] = ok ? (uint32_t)((((b * a.H + iy) * a.W + ix) * CIN + ((slot ^ ((p >> 1) & 7)) << 3)) * 2) : kOOBp;
    }
    auto issue_plane = [&](int kc) {
#pragma unroll
        for (int i = 0; i < NPL; ++i) dma16p(rsrc_x, smem + (i * NTH + wave * 64) * 16, pvoff[i], kc * 128);
    };
    const int srcchunk = (tid & 7) ^ ((tid >> 4) & 7);
    uint32_t wvoff[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i)
        wvoff[i] = (uint32_t)(((tile_n * BN + i * (NTH / 8) + (tid >> 3)) * a.Ktot + srcchunk * 8) * 2);
    // K-step t = (plane kc = t / 9, tap = t % 9); its weights are K-slice (tap * KC + kc) of [Cout][3][3][Cin]
    auto issue_w = [&](int t, int slot) {
        const int kcw = t / 9, tapw = t - kcw * 9;
#pragma unroll
        for (int i = 0; i < NBW; ++i)
            dma16p(rsrc_w, smem + WOFF + slot * WSTAGE + (i * NTH + wave * 64) * 16, wvoff[i], (tapw * KC + kcw) * 128);
    };

    f32x16_t acc[TN][TMR];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_t b4 = *(const DIR_GLOBAL f32x4_t*)(a.bias + tile_n * BN + (wn * TN + i) * 32 + 8 * g + 4 * lhi);
#pragma unroll
            for (int j = 0; j < TMR; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = b4[e];
        }

    const int wswz = (lane >> 1) & 7;
    int woffk[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) woffk[ks] = lrow * 128 + (((2 * ks + lhi) ^ wswz) << 4);

    // prologue: plane 0, then weight stages 0 and 1 (issue order matters for the counted waits below)
    issue_plane(0);
    issue_w(0, 0);
    issue_w(1, 1);
    int issued = 2;                        // weight stages issued so far
    int slot_c = 0, slot_i = 2;
    int kc = 0, tap = 0, r = 0, s = 0;     // the step being COMPUTED
    for (int t = 0; t < T; ++t) {
        // need weight stage t (and, at tap 0, this plane - issued BEFORE stage t at a plane switch, see
        // below); may leave the newest stage, t + 1, in flight
        if (issued > t + 1) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NBW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ring_barrier();      // stage t landed for every wave; slot of stage t - 1 is free
        if (issued < T) {
            issue_w(issued, slot_i);
            ++issued;
            if (++slot_i == NSTW) slot_i = 0;
        }
        const char* wst = smem + WOFF + slot_c * WSTAGE;
        frag_t xf[TMR][4];
#pragma unroll
        for (int j = 0; j < TMR; ++j) {
            const int p = (wrow * TMR + j + r) * PW + s + lrow;
            const int swz = (p >> 1) & 7;
            const char* row = smem + p * 128;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) xf[j][ks] = *(const frag_t*)(row + (((2 * ks + lhi) ^ swz) << 4));
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            frag_t wf[TN];
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = *(const frag_t*)(wst + (wn * TN + i) * 4096 + woffk[ks]);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TMR; ++j) acc[i][j] = DT::mfma32(wf[i], xf[j][ks], acc[i][j]);
        }
        if (++slot_c == NSTW) slot_c = 0;
        if (++tap == 9) {
            tap = 0;
            r = s = 0;
            ++kc;
            if (kc < KC) {
                // plane switch: every wave must be done reading the old plane before the new one may land.
                // The two weight stages already in flight (t + 1, t + 2) are OLDER than the plane request, so
                // the next step's counted wait - which may leave only the newest weight stage outstanding -
                // cannot be used as is: wait for everything once per plane instead.
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ring_barrier();
                issue_plane(kc);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else if (++s == 3) {
            s = 0;
            ++r;
        }
    }
    __syncthreads();  // plane and weight slots become epilogue staging
    Ovf<DT> ovf;

    char* ebase = smem + wave * (32 * EROW);
    constexpr int LPR = TN * 4, RPP = 64 / LPR, NPASS = 32 / RPP;
    const int ecol = (lane % LPR) * 8;
    const int erow = lane / LPR;
    const int ncol = tile_n * BN + wn * TN * 32 + ecol;
#pragma unroll
    for (int j = 0; j < TMR; ++j) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4_t v = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *(f32x4_t*)(ebase + lrow * EROW + (i * 32 + 8 * g + 4 * lhi) * 4) = v;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int oy = oy0 + wrow * TMR + j;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int mrow = pass * RPP + erow;
            const f32x4_t f0 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4);
            const f32x4_t f1 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4 + 16);
            const int ox = ox0 + mrow;
            if (oy < a.OH && ox < a.OW) {
                float v[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
                const size_t o = ((size_t)(b * a.OH + oy) * a.OW + ox) * a.Cout + ncol;
                if (a.res) {
                    const u32x4_t rv = gload16(a.res + o);
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
                gstore16(a.y + o, ov);
                ovf.see(ov);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    ovf.flush(a.ovf);
}

bool conv_patch3x3s_admissible(const ConvArgs& a) {
    return a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.H == a.OH && a.W == a.OW &&
           (a.Cin == 256 || a.Cin == 512) && a.Cout % 256 == 0;
}

template <class DT, int C>
static hipError_t launch_patch_s(const ConvArgs& a, hipStream_t stream) {
    constexpr int LDS = 6 * 512 * 16 + 3 * 256 * 128;   // one plane + three weight stages = 144 KiB
    static_assert(LDS <= 160 * 1024 && LDS >= 8 * 32 * (4 * 128 + 16), "LDS map (the staging area aliases it)");
    auto kern = conv_patch3x3s_kernel<DT, C>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    const long blocks = (long)a.B * ((a.OH + 7) / 8) * ((a.OW + 31) / 32) * (a.Cout / 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), LDS, stream, b);
    return hipGetLastError();
}

hipError_t conv_patch3x3s_launch(const ConvArgs& a, int dtype, hipStream_t stream) {
    if (a.Cin == 256)
        return dtype == DIR_BF16 ? launch_patch_s<BF16, 256>(a, stream) : launch_patch_s<FP16, 256>(a, stream);
    return dtype == DIR_BF16 ? launch_patch_s<BF16, 512>(a, stream) : launch_patch_s<FP16, 512>(a, stream);
}

}  // namespace dir
