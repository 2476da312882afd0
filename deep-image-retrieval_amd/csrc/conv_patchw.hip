// conv_patchw.hip — 3x3 stride-1 convolutions from an LDS-resident patch, 512 pixels x 128 output channels per
// workgroup, the patch one 32-CHANNEL plane at a time and double-buffered (gfx950).
//
// conv_patch.hip's plane-at-a-time kernel (256 pixels x 256 channels, 64-channel planes) was measured against two
// timing-only experiments: with half of its weight stream removed it runs 9 % faster, without
// the reload of the plane at every plane switch 5 % faster.  This form goes after both:
//   * the tile is 16 x 32 pixels x 128 channels - the same 128 accumulator registers per lane, but every weight
//     stage (fetched from L2 by every workgroup) now feeds 512 pixels instead of 256: 0.59 MB of weights + 0.31 MB
//     of patch per 302 MFLOP tile instead of 1.18 + 0.19 MB;
//   * planes are 32 channels deep (18 x 34 pixels x 64 B = 39 KB), so TWO fit: plane q + 1 is requested when plane
//     q's first weight stage starts and has three stages (144 MFMAs per wave) to land - no exposed reload;
//   * a weight stage is one filter ROW of one plane: 3 taps x 128 channels x 32 input channels = 24 KB, three slots,
//     two in flight; one barrier per stage (48 MFMAs per wave).
// LDS: 2 x 40 KB planes + 3 x 24 KB weight slots = 152 KB; the epilogue staging (8 waves x 32 x 528 B) aliases it.
// Everything else is the conv_igemm design: LDS-DMA through buffer descriptors with out-of-range zero fill, swapped
// MFMA roles (A = weights, B = pixels), bias-initialised accumulators, XOR-swizzled rows (here 64-byte rows, the four
// 16-byte chunks swizzled by (row >> 2) & 3), fp32 staging for coalesced 16-byte stores.
#include "dir_common.h"
#include "conv_igemm.h"

// Timing-only experiment builds (scripts/exp_abl.sh conv_patchw DIR_PATCHW_ABL <bits>): 1 = no global stores in the epilogue
// (staging kept), 2 = no epilogue at all (one never-taken store keeps the accumulators alive), 4 = no prologue wait: the first
// barrier does not wait for plane 0 / the first weight stage.  What a persistent form (next tile's prologue under this tile's
// epilogue) could hide at most.  Results are NOT valid convolutions.
// Round 5 - pricing a fused Winograd F(2x2, 3x3) form of this kernel BEFORE building it (profiles/r05_winograd_ablation.txt):
// 8 = the MFMA work of the Winograd-domain GEMMs (16 xi x [tiles x K] x [K x couts] per 4 output pixels = 4/9 of the direct
// conv: 22 of the 48 MFMAs of a stage, with their fragment reads), 16 = its WEIGHT stream (16 transformed taps instead of 9:
// 5 DMA instructions per stage instead of 3), 32 = its PATCH stream (the accumulators of all 16 xi do not fit the register
// file, so the separable form walks the patch once per xi ROW and a tile holds 64 channels instead of 128: 8 x the plane
// fetches per 128-channel tile).  24 / 56 = the kernel a Winograd form could at best be - its input / output transforms
// (VALU + LDS) not even counted.  64 = no LDS-DMA at all (and no waits for it): fragment reads + MFMAs + barriers + epilogue alone;
// 128 = no fragment reads / MFMAs: the DMA stream, its waits and the barriers alone.  What a loader / consumer split of this
// kernel's waves could overlap at best (profiles/r05_patchw_phases.txt).
#ifndef DIR_PATCHW_ABL
#define DIR_PATCHW_ABL 0
#endif

namespace dir {

static constexpr uint32_t kOOBw = 0x80000000u;

__device__ __forceinline__ void dma16w(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)lds, 16, voff, soff, 0, 0);
}

template <class DT>
__global__ void __launch_bounds__(512) conv_patch3x3w_kernel(const ConvArgs a) {
    constexpr int TH = 16, TW = 32, PH = TH + 2, PW = TW + 2, PP = PH * PW;   // 612 patch pixels
    constexpr int NTH = 512, BN = 128;
    constexpr int NPL = (PP * 4 + NTH - 1) / NTH;       // 5 DMA instructions per lane per plane
    constexpr int PLANE_BYTES = NPL * NTH * 16;         // 40960
    constexpr int TN = 4, TMR = 2;                      // wave tile: 128 channels x 2 output rows of 32 pixels
    constexpr int WSTAGE = 3 * BN * 64, NBW = WSTAGE / (NTH * 16);   // 24 KB, 3 instructions per lane
    constexpr int NBW_X = (DIR_PATCHW_ABL & 16) ? 2 : 0;             // (experiment) extra weight DMA instructions per stage
    constexpr int NPL_X = (DIR_PATCHW_ABL & 32) ? 7 : 0;             // (experiment) extra fetches of every plane
    constexpr int NSTW = 3;
    constexpr int WOFF = 2 * PLANE_BYTES;
    constexpr int EROW = TN * 128 + 16;
    typedef typename DT::frag_t frag_t;
    static_assert(WOFF + NSTW * WSTAGE <= 160 * 1024 && NBW * NTH * 16 == WSTAGE, "LDS map");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;

    const int NQ = a.Cin / 32;                 // planes
    const int NS = NQ * 3;                     // weight stages: (plane q, filter row r)
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

    // patch: 4 lanes per pixel (64 bytes of plane 0; plane q adds q * 64 bytes through the scalar offset)
    uint32_t pvoff[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int P = i * NTH + tid;
        const int p = P >> 2, pos = P & 3;
        const int py = p / PW, px = p - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool ok = p < PP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        pvoff[i] = ok ? (uint32_t)((((b * a.H + iy) * a.W + ix) * a.Cin + ((pos ^ ((p >> 2) & 3)) << 3)) * 2) : kOOBw;
    }
    auto issue_plane = [&](int q) {
        if (DIR_PATCHW_ABL & 64) return;
        char* dst = smem + (q & 1) * PLANE_BYTES;
#pragma unroll
        for (int i = 0; i < NPL; ++i) dma16w(rsrc_x, dst + (i * NTH + wave * 64) * 16, pvoff[i], q * 64);
#pragma unroll
        for (int rep = 0; rep < NPL_X; ++rep)       // (experiment: the same plane again - same destination, same bytes)
#pragma unroll
            for (int i = 0; i < NPL; ++i) dma16w(rsrc_x, dst + (i * NTH + wave * 64) * 16, pvoff[i], q * 64);
    };
    // weights of stage (q, r): LDS image [tap s][channel n][64 B]; linear index L = (s * 128 + n) * 4 + pos
    uint32_t wvoff[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int L = i * NTH + tid;
        const int s = L >> 9, n = (L & 511) >> 2, pos = L & 3;
        wvoff[i] = (uint32_t)((((tile_n * BN + n) * 9 + s) * a.Cin + ((pos ^ ((n >> 2) & 3)) << 3)) * 2);
    }
    auto issue_w = [&](int sigma, int slot) {
        if (DIR_PATCHW_ABL & 64) return;
        const int q = sigma / 3, r = sigma - q * 3;
        char* dst = smem + WOFF + slot * WSTAGE;
#pragma unroll
        for (int i = 0; i < NBW; ++i) dma16w(rsrc_w, dst + (i * NTH + wave * 64) * 16, wvoff[i], (r * 3 * a.Cin + q * 32) * 2);
#pragma unroll
        for (int i = 0; i < NBW_X; ++i)             // (experiment: 16 / 9 of the weight bytes, from the neighbouring filter row)
            dma16w(rsrc_w, dst + (i * NTH + wave * 64) * 16, wvoff[i], (((r + 1) % 3) * 3 * a.Cin + q * 32) * 2);
    };

    f32x16_t acc[TN][TMR];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_t b4 = *(const DIR_GLOBAL f32x4_t*)(a.bias + tile_n * BN + i * 32 + 8 * g + 4 * lhi);
#pragma unroll
            for (int j = 0; j < TMR; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = b4[e];
        }

    // A fragment of channel tile i, K-sub-step kk: row n = i * 32 + lrow of a tap's [128][64 B] block
    int woffk[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) woffk[kk] = lrow * 64 + (((2 * kk + lhi) ^ ((lrow >> 2) & 3)) << 4);

    // prologue: plane 0, weight stages 0 and 1 - in this order (the counted waits below rely on it)
    issue_plane(0);
    issue_w(0, 0);
    issue_w(1, 1);
    int q = 0, r = 0;          // the stage being computed
    int slot_c = 0;
    bool plane_before = false;   // the previous iteration requested a plane (5 more operations in flight)
    for (int sigma = 0; sigma < NS; ++sigma) {
        // need: weight stage sigma (and, at r == 0, plane q - requested three stages ago, before stage sigma).
        // May stay in flight: what the PREVIOUS iteration requested - [plane q + 1,] weight stage sigma + 1.
        if (((DIR_PATCHW_ABL & 4) && sigma == 0) || (DIR_PATCHW_ABL & 64)) {
            // (experiment: skip the wait for the prologue's loads)
        } else if (sigma + 1 >= NS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (plane_before) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPL * (1 + NPL_X) + NBW + NBW_X) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NBW + NBW_X) : "memory");
        }
        ring_barrier();   // stage sigma (and plane q) landed everywhere; everyone is past stage sigma - 1
        plane_before = false;
        if (r == 0 && q + 1 < NQ) {     // the other plane buffer was last read during plane q - 1
            issue_plane(q + 1);
            plane_before = true;
        }
        if (sigma + 2 < NS) {
            int slot_n = slot_c + 2;
            if (slot_n >= NSTW) slot_n -= NSTW;
            issue_w(sigma + 2, slot_n);
        }
        const char* plane = smem + (q & 1) * PLANE_BYTES;
        const char* wst = smem + WOFF + slot_c * WSTAGE;
#pragma unroll
        for (int s = 0; s < ((DIR_PATCHW_ABL & 128) ? 0 : 3); ++s) {
            frag_t xf[TMR][2], wf[2][TN];
#pragma unroll
            for (int j = 0; j < TMR; ++j) {
                const int p = (wave * TMR + j + r) * PW + s + lrow;
                const int swz = (p >> 2) & 3;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) xf[j][kk] = *(const frag_t*)(plane + p * 64 + (((2 * kk + lhi) ^ swz) << 4));
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < TN; ++i) wf[kk][i] = *(const frag_t*)(wst + s * (BN * 64) + i * 2048 + woffk[kk]);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TMR; ++j) {
                        // (experiment, bit 8: 22 of the 48 MFMAs of a stage - tap 0 whole, tap 1's first K half on three
                        // channel tiles - and only their fragment reads survive)
                        if ((DIR_PATCHW_ABL & 8) && !(s == 0 || (s == 1 && kk == 0 && i < 3))) continue;
                        acc[i][j] = DT::mfma32(wf[kk][i], xf[j][kk], acc[i][j]);
                    }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this stage's LDS reads retired before the next barrier
        if (++slot_c == NSTW) slot_c = 0;
        if (++r == 3) {
            r = 0;
            ++q;
        }
    }
    __syncthreads();  // planes and weight slots become epilogue staging
    Ovf<DT> ovf;
    if (DIR_PATCHW_ABL & 2) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TMR; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
        if (sum == 1.2345678e30f) a.y[0] = 1;
        return;
    }

    char* ebase = smem + wave * (32 * EROW);
    constexpr int LPR = TN * 4, RPP = 64 / LPR, NPASS = 32 / RPP;
    const int ecol = (lane % LPR) * 8;
    const int erow = lane / LPR;
    const int ncol = tile_n * BN + ecol;
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
        const int oy = oy0 + wave * TMR + j;
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
                if (!(DIR_PATCHW_ABL & 1) || ov[0] == 0x12345678u) gstore16(a.y + o, ov);
                ovf.see(ov);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    ovf.flush(a.ovf);
}

// ---- loader / consumer form (round 5) ---------------------------------------------------------------------------------
// Phases compiled out of the kernel above (profiles/r05_patchw_phases.txt, layer3's conv2 at batch 32): fragment reads + MFMAs +
// epilogue alone 103 us, the LDS-DMA stream + its waits alone 51 us, together 140 us - when the waves that issue the next
// stage are the waves that multiply this one, the two phases barely overlap (a DMA instruction holds its wave at issue while the
// CU's request queue is busy: round-3 finding, conv_ring.hip / sim_split_lc_kernel).  Same tile, stages, LDS map, fragment
// reads and MFMA order here - bit-identical results - with the work split by wave role: TWELVE waves, waves 0-7 only read
// fragments and multiply (two per SIMD), waves 8-11 only issue LDS-DMA and wait for it (one per SIMD; 10 instructions per plane
// and 6 per weight stage each, a counted vmcnt over one kind of op).  One barrier per stage is the hand-off in both directions.
// (s_setprio on either role - consumers 1 / 3, loaders 1 - measured: 14.11-14.18 ms per step in every combination, 14.13-14.15
// without; not kept.)
// Twelve waves = three per SIMD = 168 VGPRs per wave: the consumers' 128 accumulator registers leave 40, so fragments are
// fetched one K half (kk) at a time.
template <class DT>
__global__ void __launch_bounds__(768) conv_patch3x3w_lc_kernel(const ConvArgs a) {
    constexpr int TH = 16, TW = 32, PH = TH + 2, PW = TW + 2, PP = PH * PW;   // 612 patch pixels
    constexpr int NLD = 256, BN = 128;                    // loader lanes
    constexpr int NPL = (PP * 4 + NLD - 1) / NLD;         // 10 DMA instructions per loader lane per plane
    constexpr int PLANE_BYTES = NPL * NLD * 16;           // 40960
    constexpr int TN = 4, TMR = 2;
    constexpr int WSTAGE = 3 * BN * 64, NBW = WSTAGE / (NLD * 16);   // 24 KB, 6 instructions per loader lane
    constexpr int NSTW = 3;
    constexpr int WOFF = 2 * PLANE_BYTES;
    constexpr int EROW = TN * 128 + 16;
    typedef typename DT::frag_t frag_t;
    static_assert(WOFF + NSTW * WSTAGE <= 160 * 1024 && NBW * NLD * 16 == WSTAGE, "LDS map");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;

    const int NQ = a.Cin / 32;                 // planes
    const int NS = NQ * 3;                     // weight stages: (plane q, filter row r)
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

    if (wave >= 8) {
        // ================================ loaders ==============================================================================
        const int lt = tid - 512;              // 0 .. 255
        const int lw = wave - 8;
        const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
        uint32_t pvoff[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int P = i * NLD + lt;
            const int p = P >> 2, pos = P & 3;
            const int py = p / PW, px = p - py * PW;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
            const bool ok = p < PP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            pvoff[i] = ok ? (uint32_t)((((b * a.H + iy) * a.W + ix) * a.Cin + ((pos ^ ((p >> 2) & 3)) << 3)) * 2) : kOOBw;
        }
        // a.w_pw (conv_patch3x3w_pack): the filter as the sequence of 24 KB LDS stage images - a DMA instruction copies one
        // contiguous KB instead of gathering sixteen 64-byte runs of the [Cout][3][3][Cin] layout
        const bool packed = a.w_pw != nullptr;
        const __amdgpu_buffer_rsrc_t rsrc_wp = __builtin_amdgcn_make_buffer_rsrc((void*)(packed ? a.w_pw : a.w), 0, a.w_bytes, 0x00020000);
        uint32_t wvoff[NBW];
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int L = i * NLD + lt;
            const int s = L >> 9, n = (L & 511) >> 2, pos = L & 3;
            wvoff[i] = packed ? (uint32_t)(L * 16) : (uint32_t)((((tile_n * BN + n) * 9 + s) * a.Cin + ((pos ^ ((n >> 2) & 3)) << 3)) * 2);
        }
        auto issue_plane = [&](int q) {
            char* dst = smem + (q & 1) * PLANE_BYTES;
#pragma unroll
            for (int i = 0; i < NPL; ++i) dma16w(rsrc_x, dst + (i * NLD + lw * 64) * 16, pvoff[i], q * 64);
        };
        auto issue_w = [&](int sigma, int slot) {
            const int q = sigma / 3, r = sigma - q * 3;
            char* dst = smem + WOFF + slot * WSTAGE;
            const int soff = packed ? (tile_n * NS + sigma) * WSTAGE : (r * 3 * a.Cin + q * 32) * 2;
#pragma unroll
            for (int i = 0; i < NBW; ++i) dma16w(rsrc_wp, dst + (i * NLD + lw * 64) * 16, wvoff[i], soff);
        };
        // prologue: plane 0, weight stages 0 and 1 - in this order (the counted waits below rely on it)
        issue_plane(0);
        issue_w(0, 0);
        if (NS > 1) issue_w(1, 1);
        int q = 0, r = 0, slot_c = 0;
        bool plane_before = false;
        for (int sigma = 0; sigma < NS; ++sigma) {
            // landed: weight stage sigma (and, at r == 0, plane q - requested three stages ago).  May stay in flight: what the
            // previous iteration requested - [plane q + 1,] weight stage sigma + 1.  Only LDS-DMA enters this wave's queue.
            if (sigma + 1 >= NS) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (plane_before) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPL + NBW) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NBW) : "memory");
            }
            ring_barrier();   // hand-off: stage sigma (and plane q) complete; the consumers have left stage sigma - 1
            plane_before = false;
            if (r == 0 && q + 1 < NQ) {     // the other plane buffer was last read during plane q - 1
                issue_plane(q + 1);
                plane_before = true;
            }
            if (sigma + 2 < NS) {
                int slot_n = slot_c + 2;
                if (slot_n >= NSTW) slot_n -= NSTW;
                issue_w(sigma + 2, slot_n);
            }
            if (++slot_c == NSTW) slot_c = 0;
            if (++r == 3) {
                r = 0;
                ++q;
            }
        }
        ring_barrier();   // (the consumers' barrier before the epilogue staging takes the ring's place)
        return;
    }

    // ==================================== consumers ================================================================================
    f32x16_t acc[TN][TMR];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_t b4 = *(const DIR_GLOBAL f32x4_t*)(a.bias + tile_n * BN + i * 32 + 8 * g + 4 * lhi);
#pragma unroll
            for (int j = 0; j < TMR; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = b4[e];
        }
    int woffk[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) woffk[kk] = lrow * 64 + (((2 * kk + lhi) ^ ((lrow >> 2) & 3)) << 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the bias: the only VMEM loads of a consumer before the epilogue

    int q = 0, r = 0, slot_c = 0;
    for (int sigma = 0; sigma < NS; ++sigma) {
        ring_barrier();
        const char* plane = smem + (q & 1) * PLANE_BYTES;
        const char* wst = smem + WOFF + slot_c * WSTAGE;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            // same (kk, i, j) MFMA order per accumulator as conv_patch3x3w_kernel: bit-identical sums
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                frag_t xf[TMR], wf[TN];
#pragma unroll
                for (int j = 0; j < TMR; ++j) {
                    const int p = (wave * TMR + j + r) * PW + s + lrow;
                    xf[j] = *(const frag_t*)(plane + p * 64 + (((2 * kk + lhi) ^ ((p >> 2) & 3)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) wf[i] = *(const frag_t*)(wst + s * (BN * 64) + i * 2048 + woffk[kk]);
                // Issue order: the PIXEL fragment is held over four consecutive MFMAs while the weight fragment changes (j outer, i
                // inner - and written so, because hipcc's own schedule of the i-outer loop interleaved them irregularly).  The kernel
                // is power-bound (MfmaUtil x clock is constant across its forms); which operand toggles between consecutive
                // MFMAs is worth 2-3 % of the launch: 126.5 -> 123.5 us in layer3, 117.5 -> 113.5 us in layer4, step 14.14 -> 14.07 ms
                // (A/B, three orders, profiles/r05_patchw_phases.txt).  Per accumulator the k order is unchanged: bit-identical.
#pragma unroll
                for (int j = 0; j < TMR; ++j)
#pragma unroll
                    for (int i = 0; i < TN; ++i) acc[i][j] = DT::mfma32(wf[i], xf[j], acc[i][j]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this stage's LDS reads retired before the next barrier
        if (++slot_c == NSTW) slot_c = 0;
        if (++r == 3) {
            r = 0;
            ++q;
        }
    }
    ring_barrier();  // planes and weight slots become epilogue staging (the loaders leave here)
    Ovf<DT> ovf;
    char* ebase = smem + wave * (32 * EROW);
    constexpr int LPR = TN * 4, RPP = 64 / LPR, NPASS = 32 / RPP;
    const int ecol = (lane % LPR) * 8;
    const int erow = lane / LPR;
    const int ncol = tile_n * BN + ecol;
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
        const int oy = oy0 + wave * TMR + j;
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

bool conv_patch3x3w_admissible(const ConvArgs& a) {
    return a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.H == a.OH && a.W == a.OW && a.Cin % 32 == 0 &&
           a.Cin >= 64 && a.Cout % 128 == 0 && (size_t)a.B * a.H * a.W * a.Cin * 2 < (1ull << 31) &&
           (size_t)a.Cout * a.Ktot * 2 < (1ull << 31);
}

// The filter as conv_patch3x3w_lc_kernel's LDS stage images: 16-byte piece ((tile_n * NQ + q) * 3 + r) * 1536 + L, L = (s * 128 + n) * 4 + pos,
// = w[tile_n * 128 + n][r][s][q * 32 + (pos ^ ((n >> 2) & 3)) * 8 .. + 8]  (same bytes, same size).
__global__ void __launch_bounds__(256) pack_patchw_kernel(const uint16_t* w, uint16_t* out, int Cout, int Cin) {
    const int NQ = Cin / 32;
    const long pieces = (long)Cout * 9 * Cin / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (long)gridDim.x * 256) {
        const int L = (int)(i % 1536);
        long t = i / 1536;
        const int r = (int)(t % 3);
        t /= 3;
        const int q = (int)(t % NQ);
        const int tn = (int)(t / NQ);
        const int s = L >> 9, n = (L & 511) >> 2, pos = L & 3;
        const size_t src = ((size_t)((tn * 128 + n) * 9 + r * 3 + s)) * Cin + q * 32 + ((pos ^ ((n >> 2) & 3)) << 3);
        gstore16(out + i * 8, gload16(w + src));
    }
}

hipError_t conv_patch3x3w_pack(const uint16_t* w, uint16_t* out, int Cout, int Cin, hipStream_t stream) {
    const long pieces = (long)Cout * 9 * Cin / 8;
    const int grid = (int)((pieces + 255) / 256 < 1024 ? (pieces + 255) / 256 : 1024);
    hipLaunchKernelGGL(pack_patchw_kernel, dim3(grid), dim3(256), 0, stream, w, out, Cout, Cin);
    return hipGetLastError();
}

template <class DT>
static hipError_t launch_patch_w(const ConvArgs& a, hipStream_t stream) {
    constexpr int LDS = 2 * 5 * 512 * 16 + 3 * 3 * 128 * 64;   // two planes + three weight stages = 152 KiB
    static_assert(LDS <= 160 * 1024 && LDS >= 8 * 32 * (4 * 128 + 16), "LDS map (the staging area aliases it)");
    ConvArgs b = a;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    const long blocks = (long)a.B * ((a.OH + 15) / 16) * ((a.OW + 31) / 32) * (a.Cout / 128);
    if (!env().no_patchw_lc) {   // the loader / consumer form (twelve waves); DIRTORCH_AMD_NO_PATCHW_LC = the one-role kernel
        void* scratch = nullptr;
        if (!b.w_pw && env().patchw_pack) {   // DIRTORCH_AMD_PATCHW_PACK: a per-op launch (no packed copy kept anywhere) packs into stream-ordered scratch
            if (hipError_t e = hipMallocAsync(&scratch, b.w_bytes, stream); e != hipSuccess) return e;
            if (hipError_t e = conv_patch3x3w_pack(a.w, (uint16_t*)scratch, a.Cout, a.Cin, stream); e != hipSuccess) {
                (void)hipFreeAsync(scratch, stream);
                return e;
            }
            b.w_pw = (const uint16_t*)scratch;
        }
        auto kern = conv_patch3x3w_lc_kernel<DT>;
        static std::atomic<uint64_t> attr_lc{0};
        if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_lc); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(768), LDS, stream, b);
        hipError_t e = hipGetLastError();
        if (scratch) {
            const hipError_t f = hipFreeAsync(scratch, stream);
            if (e == hipSuccess) e = f;
        }
        return e;
    }
    auto kern = conv_patch3x3w_kernel<DT>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), LDS, stream, b);
    return hipGetLastError();
}

hipError_t conv_patch3x3w_launch(const ConvArgs& a, int dtype, hipStream_t stream) {
    return dtype == DIR_BF16 ? launch_patch_w<BF16>(a, stream) : launch_patch_w<FP16>(a, stream);
}

}  // namespace dir
