// stem_u8.hip — the DIR_FP16P stem on the RAW uint8 image (gfx950): ToTensor + Normalize + 7x7 s2 conv + BN + ReLU +
// 3x3 s2 max-pool (dirtorch/utils/transforms.py:617-623, dirtorch/nets/backbones/resnet.py:115-119,158-161).
//
// The reference's real input is a uint8 image; the integers 0 ... 255 are EXACT in one fp16 plane, so on this feed the
// image needs no lo plane and the paired stem's third product (w_hi . x_lo) has nothing to multiply.  The affine
// normalisation moves into the filter and the bias:
//     y = b + sum_{taps inside the image} w . ((u / 255 - mean_c) / std_c)
//       = [b - sum_{taps inside} w . mean_c / std_c]  +  sum_{all taps} [w . 256 / (255 std_c)] . (u / 256)
// (a tap outside the image is ZERO in normalised space - zero padding comes after Normalize - and u = 0 there, so the
// second sum needs no mask; the first does: the folded bias is exact for interior pixels and a per-border-class
// correction table, 6 row classes x 6 column classes x 64 channels, restores the taps the three padded rows / columns
// drop.)  x = u / 256 is exact in fp16, w' = w . 256 / (255 std_c) is an fp16 PAIR (~22 bits), every product w'_hi . x
// is exact in fp32: two MFMAs per term instead of three, half the patch stream, at no precision cost.
//
//   prep_input_u8_kernel   uint8 NHWC -> 2x2 space-to-depth plane [B, H2, W2, 16] of u / 256 (12 real + 4 zero channels)
//   stem_pool_u8_kernel    persistent, one 8-wave workgroup per CU.  What differs from conv_pair.hip's paired stem:
//     * max-pool in REGISTERS: the MFMA accumulator layout puts conv column l in lane l, so the horizontal 3-max is two
//       whole-wave DPP shifts (wave_shr:1 / wave_shl:1, scripts/probes/dpp_wave_shift.hip) per value and the vertical one a
//       max of the wave's own two conv rows; only the row that straddles two waves goes through LDS (32 KB per tile
//       instead of a 64 KB fp32 conv tile written once and read nine times);
//     * tiles walk DOWN a column strip in segments of `seg_rows` pooled rows; the pool row that straddles two TILES is
//       carried (wave 0 of tile t + 1 finishes the row wave 3 of tile t began), so 8 conv rows yield 4 pooled rows
//       where independent 3 x 15 tiles recompute two of every eight rows;
//     * the pooled rows of tile t are split into (hi, lo) and stored while tile t + 1 multiplies (exchange buffers
//       double, patch buffers triple): ONE barrier per tile.
// Layouts (patch planes, filter fragments straight in MFMA operand layout, swapped MFMA roles) are conv_pair.hip's.
#include "dir_common.h"
#include "conv_igemm.h"
#include "pointwise.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

// Timing-only experiment builds (scripts/exp_abl.sh stem_u8 DIR_STEMU8_ABL <bits>): 1 = no MFMAs / fragment reads,
// 2 = no emit phase (no pooled stores), 4 = no patch DMA.  NOT valid results.
#ifndef DIR_STEMU8_ABL
#define DIR_STEMU8_ABL 0
#endif

namespace dir {

static constexpr uint32_t kOOBu = 0x80000000u;

__device__ __forceinline__ void dma16u(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)lds, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ void split2u(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = FP16::pack(a, b);
    float ha, hb;
    FP16::unpack(hi, ha, hb);
    lo = FP16::pack(a - ha, b - hb);
}
// value of the lane to the left / right (whole-wave shift by one lane; lane 0 / lane 63 read 0 - both belong to even
// columns, whose 3-max nobody uses).  old = 0 + bound_ctrl: no tied operand, so no register copies around the DPP move.
__device__ __forceinline__ float lane_left(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_right(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

struct StemU8Args {
    const uint8_t* img;         // RAW form: the uint8 NHWC image itself [B, H, W, 3] (W even)
    uint32_t img_bytes;
    const uint16_t* x;          // s2d plane [B, H2, W2, 16], u / 256 (XPAIR: the hi plane of the normalised image pair)
    const uint16_t* xl;         // XPAIR: its lo plane
    const uint16_t *wh, *wl;    // folded filter pair [64][4][4][16]
    const float* bias;          // [64] folded bias (all 147 taps inside the image); added - with the ReLU - to the POOLED values
    const float* corr;          // [6][6][64] border-class corrections (class 0 = interior = zeros)
    uint16_t *yh, *yl;          // pooled pair [B, PH, PW, 64]
    int B, H, W, H2, W2, OH, OW, PH, PW;
    int tiles_x, nseg, seg_rows, nitems;
    uint32_t x_bytes;
    int* ovf;
};

// one 8 x 32 conv tile of a segment
struct TileU8 {
    int valid;
    int item, t, nt;     // work item, tile index inside its segment, tiles of the segment
    int b, pw0;          // image, first pooled column
    int pbase, pe;       // pooled row of k = 0, end of the segment's pooled rows
    int c;               // first conv row
};

// border class of conv row / column o of an axis of length n (n >= 7): which taps r = 0..6 (input 2 o - 3 + r) lie inside.
// 0: all; 1: r >= 3 (o = 0); 2: r >= 1 (o = 1); 3 / 4 / 5: r <= 5 / 4 / 3 (the last one or two outputs)
__device__ __forceinline__ int border_class(int o, int n) {
    const int t = 2 * o + 4 - n;
    return o == 0 ? 1 : (o == 1 ? 2 : (t <= 0 ? 0 : 2 + t));
}

// NRP = conv row PAIRS per workgroup (= pooled rows per tile): 4 -> 8 waves, 8 x 32 conv tiles, one workgroup per CU;
// 2 -> 4 waves, 4 x 32 conv tiles, TWO workgroups per CU - their barriers are independent, so one's matrix phase runs
// under the other's pooling / emit phase on the same SIMDs (the phases of one workgroup's waves move in lock-step).
// RAW: the patch comes straight from the uint8 NHWC image - every thread fetches the 2 x 6 bytes of ITS space-to-depth pixel two
// tiles ahead (six 2-byte buffer loads into registers; needs an even W: rows of 3 W bytes then keep 6-byte groups 2-byte aligned),
// converts them to u / 256 one tile later and writes the two 16-byte plane chunks the MFMA fragments read - prep_input_u8's
// arithmetic, without the launch, without the 32-bytes-per-pixel plane in HBM (3 bytes per image pixel in instead of 8 + 8 out + in).
// XPAIR (never with RAW): the generic paired stem of DIR_FP16P on this kernel's structure - the image is an fp16 PAIR (any fp32
// input: the reference's normalised tensor, prep_input_pair's two planes), nothing is folded (a.bias is bn1's, there is no border
// table: zero padding of a normalised image is the plain OOB zero), every term is three MFMAs in conv_pair.hip's order
// (w_hi.x_hi, w_hi.x_lo, w_lo.x_hi) from accumulators that START at the bias - the sums stem_pool_pair_persist_kernel forms, bit for
// bit - and the patch pair is double-buffered with the next tile's four DMA pieces issued one phase ahead (2 x 16 + 32 KB: two
// workgroups per CU).
template <int NRP, bool RAW, bool XPAIR = false>
__global__ void __launch_bounds__(128 * NRP, NRP == 2 ? 2 : 1) stem_pool_u8_kernel(const StemU8Args a) {
    // RAW && XPAIR: the patch pair comes straight from the fp32 NCHW image (a.img: six 8-byte loads per thread - the two horizontal
    // neighbours of one channel and row -, split into (hi, lo) one tile later): prep_input_pair's arithmetic without its launch.
    typedef FP16 DT;
    typedef DT::frag_t frag_t;
    constexpr int PTW = 15;
    constexpr int NT = 128 * NRP;               // threads
    constexpr int TH = 2 * NRP, TW = 32;
    constexpr int QW = TW + 3;                  // patch width 35, height TH + 3
    constexpr int QP = (TH + 3) * QW;           // 385 / 245 patch pixels
    constexpr int PLANE = NT * 16;              // one channel-half plane (8 channels x NT pixel slots)
    constexpr int PATCH = (XPAIR ? 4 : 2) * PLANE;   // 16 / 8 KiB (XPAIR: hi planes, then lo planes)
    constexpr int NPB = XPAIR ? 2 : 3;          // patch buffers (16 KB each for pairs: two, so that two workgroups fit a CU)
    constexpr int XROW = 16 * 256;              // one exchanged row: 16 pooled columns x 64 channels fp32
    constexpr int XBUF = 2 * NRP * XROW;        // slots 0..NRP-1: M[k] = max of conv rows 2k, 2k+1; NRP..2NRP-2: H[k], k = 1..NRP-1 (conv row 2k); 2NRP-1: KP
    static_assert(QP <= NT, "one DMA instruction per plane covers the patch");
    constexpr int X_OFF = NPB * PATCH;
    constexpr int BIAS_OFF = X_OFF + 2 * XBUF;
    constexpr int CORR_OFF = BIAS_OFF + 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int ci = wave & 1, rp = wave >> 1;     // channel tile, conv row pair

    const __amdgpu_buffer_rsrc_t rsrc_x = RAW ? __builtin_amdgcn_make_buffer_rsrc((void*)a.img, 0, a.img_bytes, 0x00020000)
                                              : __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_xl = __builtin_amdgcn_make_buffer_rsrc((void*)(XPAIR ? a.xl : a.x), 0, a.x_bytes, 0x00020000);

    // ---- the filter pair of this wave's channel tile, straight in MFMA operand layout (128 VGPRs, fetched once) ----------
    frag_t wfh[4][4], wfl[4][4];
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const size_t o = (size_t)(ci * 32 + lrow) * 256 + R * 64 + ks * 16 + lhi * 8;
            wfh[R][ks] = __builtin_bit_cast(frag_t, gload16(a.wh + o));
            wfl[R][ks] = __builtin_bit_cast(frag_t, gload16(a.wl + o));
        }
    float* lbias = (float*)(smem + BIAS_OFF);
    float* lcorr = (float*)(smem + CORR_OFF);
    if (tid < 64) lbias[tid] = a.bias[tid];
    if (!XPAIR)
        for (int i = tid; i < 36 * 64; i += NT) lcorr[i] = a.corr[i];

    // ---- tile sequence of this workgroup: items blockIdx.x, + gridDim.x, ...; item = (segment, image, column strip) -------
    auto decode = [&](TileU8& d) {
        d.valid = d.item < a.nitems;
        if (!d.valid) return;
        int r = d.item;
        const int tx = r % a.tiles_x;
        r /= a.tiles_x;
        d.b = r % a.B;
        const int seg = r / a.B;
        const int ps = seg * a.seg_rows;
        d.pe = min(a.PH, ps + a.seg_rows);
        d.nt = (2 * (d.pe - ps) + TH) / TH;      // conv rows 2 ps - 1 ... 2 pe - 1 (2 n + 1 of them) in tiles of TH
        d.pw0 = tx * PTW;
        d.pbase = ps + NRP * d.t;
        d.c = 2 * ps - 1 + TH * d.t;
    };
    auto advance = [&](TileU8& d) {
        if (!d.valid) return;
        if (++d.t < d.nt) {
            d.pbase += NRP;
            d.c += TH;
        } else {
            d.item += gridDim.x;
            d.t = 0;
            decode(d);
        }
    };
    // this thread carries patch pixel (ppy, ppx) of every tile, both channel halves: two LDS-DMA instructions per tile
    const int ppy = tid / QW, ppx = tid - ppy * QW;
    const int ppoff = (ppy * a.W2 + ppx) * 32;
    auto issue_patch = [&](const TileU8& d, char* dst) {
        const int iy = d.c - 2 + ppy, ix = 2 * d.pw0 - 3 + ppx;
        const bool ok = tid < QP && (unsigned)iy < (unsigned)a.H2 && (unsigned)ix < (unsigned)a.W2;
        const int base = ((d.b * a.H2 + d.c - 2) * a.W2 + 2 * d.pw0 - 3) * 32;       // (wave-uniform)
        const uint32_t v = ok ? (uint32_t)(base + ppoff) : kOOBu;
        if (DIR_STEMU8_ABL & 4) return;
        dma16u(rsrc_x, dst + (wave * 64) * 16, v);
        dma16u(rsrc_x, dst + (NT + wave * 64) * 16, ok ? v + 16 : kOOBu);
        if (XPAIR) {
            dma16u(rsrc_xl, dst + (2 * NT + wave * 64) * 16, v);
            dma16u(rsrc_xl, dst + (3 * NT + wave * 64) * 16, ok ? v + 16 : kOOBu);
        }
    };

    // RAW: the 12 image bytes of this thread's patch pixel, as six zero-extended 16-bit loads (bytes 2k, 2k + 1 of row 0, then row 1)
    uint32_t rw[XPAIR ? 12 : 6] = {};
    auto load_raw = [&](const TileU8& d) {
        const int iy = d.c - 2 + ppy, ix = 2 * d.pw0 - 3 + ppx;
        const bool ok = tid < QP && (unsigned)iy < (unsigned)a.H2 && (unsigned)ix < (unsigned)a.W2;
        if (XPAIR) {   // fp32 NCHW: floats (c, 2 iy + dy, 2 ix) and (.., 2 ix + 1) as one 8-byte load; W even keeps them 8-byte aligned
            const int plane = a.H * a.W;
            const uint32_t o0 = ok ? (uint32_t)((((d.b * 3) * a.H + 2 * iy) * a.W + 2 * ix) * 4) : kOOBu;
            const bool row1 = ok && 2 * iy + 1 < a.H;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const uint32_t o = (dy == 0 ? ok : row1) ? o0 + (uint32_t)((c * plane + dy * a.W) * 4) : kOOBu;
                    const u32x2_t v = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rsrc_x, o, 0, 0));
                    rw[(c * 2 + dy) * 2] = v[0];
                    rw[(c * 2 + dy) * 2 + 1] = v[1];
                }
            return;
        }
        const uint32_t o0 = ok ? (uint32_t)(((d.b * a.H + 2 * iy) * a.W + 2 * ix) * 3) : kOOBu;      // offsets >= 2^31 read as 0
        const uint32_t o1 = (ok && 2 * iy + 1 < a.H) ? o0 + (uint32_t)(a.W * 3) : kOOBu;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rw[k] = __builtin_amdgcn_raw_buffer_load_b16(rsrc_x, o0 + 2 * k, 0, 0);
            rw[3 + k] = __builtin_amdgcn_raw_buffer_load_b16(rsrc_x, o1 + 2 * k, 0, 0);
        }
    };
    auto store_raw = [&](char* dst) {   // channel (dy * 2 + dx) * 3 + c = byte 6 dy + 3 dx + c of the twelve; u / 256 is exact in fp16
        if (XPAIR) {   // (hi, lo) = (fp16(v), fp16(v - hi)) per value, channels in s2d order, 12..15 zero: prep_input_pair's planes
            float f[12];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) f[(dy * 2 + dx) * 3 + c] = __builtin_bit_cast(float, rw[(c * 2 + dy) * 2 + dx]);
            u32x4_t h0, h1, l0, l1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t hh, ll;
                split2u(f[2 * e], f[2 * e + 1], hh, ll);
                h0[e] = hh;
                l0[e] = ll;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                uint32_t hh, ll;
                split2u(f[8 + 2 * e], f[9 + 2 * e], hh, ll);
                h1[e] = hh;
                l1[e] = ll;
            }
            h1[2] = h1[3] = l1[2] = l1[3] = 0;
            *(u32x4_t*)(dst + tid * 16) = h0;
            *(u32x4_t*)(dst + (NT + tid) * 16) = h1;
            *(u32x4_t*)(dst + (2 * NT + tid) * 16) = l0;
            *(u32x4_t*)(dst + (3 * NT + tid) * 16) = l1;
            return;
        }
        float f[12];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            f[2 * k] = (float)(rw[k] & 0xffu) * 0.00390625f;
            f[2 * k + 1] = (float)((rw[k] >> 8) & 0xffu) * 0.00390625f;
        }
        u32x4_t p0, p1;
#pragma unroll
        for (int e = 0; e < 4; ++e) p0[e] = FP16::pack(f[2 * e], f[2 * e + 1]);
        p1[0] = FP16::pack(f[8], f[9]);
        p1[1] = FP16::pack(f[10], f[11]);
        p1[2] = p1[3] = 0;
        *(u32x4_t*)(dst + tid * 16) = p0;
        *(u32x4_t*)(dst + (NT + tid) * 16) = p1;
    };

    TileU8 cur, nxt, pre, prev;
    cur.item = blockIdx.x;
    cur.t = 0;
    decode(cur);
    if (!cur.valid) return;
    nxt = cur;
    advance(nxt);
    pre = nxt;
    advance(pre);
    prev.valid = 0;
    prev.t = prev.b = prev.pw0 = prev.pbase = prev.pe = 0;

    Ovf<DT> ovf;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // filter in registers, tables in LDS, before the counted waits
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {   // pin: the waits for these loads must not be re-executed inside the loop
            asm volatile("" : "+v"(wfh[R][ks]));
            asm volatile("" : "+v"(wfl[R][ks]));
        }
    if (RAW) {   // patch 0 into its buffer, patch 1 into the registers (phase 0 stores it)
        load_raw(cur);
        store_raw(smem);
        if (nxt.valid) load_raw(nxt);
    } else {
        issue_patch(cur, smem);
        if (!XPAIR && nxt.valid) issue_patch(nxt, smem + PATCH);
    }

    // phase n: [wait patch n | barrier] issue patch n + 2 -> emit the pooled rows of tile n - 1 -> multiply tile n -> pool in
    // registers -> publish the rows other waves / the next tile finish.  The last phase only emits.
    int pb = 0;       // patch buffer of `cur` (n % 3)
    int xb = 0;       // exchange buffer `cur` publishes into (n & 1)
    for (;;) {
        if (RAW)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (patch n was WRITTEN by this workgroup's own ds_writes in phase n - 1)
        else if (XPAIR)      // (the DMA form of the pair: RAW was taken above)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // patch n (issued one phase ago) landed; so did phase n - 1's stores
        else if (nxt.valid)
            asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");   // patch n landed; the 2 DMA ops of patch n + 1 may fly
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        ring_barrier();   // patch n visible; rows published in phase n - 1 visible; phase n - 1's LDS reads retired everywhere

        if (RAW) {
            // the registers hold patch n + 1 (fetched in phase n - 1): into buffer (n + 1) % NPB, which the MFMAs of phase n - 1 (or
            // n - 2) read last - every wave left them before this barrier; then the fetch of patch n + 2 goes out, a whole phase
            // ahead of its conversion
            if (nxt.valid) store_raw(smem + (pb == NPB - 1 ? 0 : pb + 1) * PATCH);
            if (pre.valid) load_raw(pre);
        } else if (XPAIR) {      // two buffers: patch n + 1 into the one the MFMAs of phase n - 1 read last
            if (nxt.valid) issue_patch(nxt, smem + (pb ^ 1) * PATCH);
        } else if (pre.valid) {  // buffer (n + 2) % 3 = (n - 1) % 3 was last read by the MFMAs of phase n - 1
            const int nb = pb == 0 ? 2 : pb - 1;
            issue_patch(pre, smem + nb * PATCH);
        }

        // ---- emit tile n - 1: rows k = 0..2 -> max(M[k], H[k + 1]); row 3 -> KP (the row tile n - 2 began), if any -------
        if (prev.valid && !(DIR_STEMU8_ABL & 2)) {
            const char* X = smem + X_OFF + (xb ^ 1) * XBUF;
            const int r4 = tid >> 7, px = (tid >> 3) & 15, c8 = tid & 7;      // r4 < NRP - 1: row k = r4; r4 = NRP - 1: the carried row
            const int ph = r4 < NRP - 1 ? prev.pbase + r4 : prev.pbase - 1;
            const int pw = prev.pw0 + px;
            const bool live = px < PTW && pw < a.PW && ph < prev.pe && (r4 < NRP - 1 || prev.t > 0);
            if (live) {
                const int o0 = px * 256 + (((2 * c8) ^ (px & 7)) << 4), o1 = px * 256 + (((2 * c8 + 1) ^ (px & 7)) << 4);
                f32x4_t m0, m1;
                if (r4 < NRP - 1) {
                    const f32x4_t a0 = *(const f32x4_t*)(X + r4 * XROW + o0), a1 = *(const f32x4_t*)(X + r4 * XROW + o1);
                    const f32x4_t b0 = *(const f32x4_t*)(X + (NRP + r4) * XROW + o0), b1 = *(const f32x4_t*)(X + (NRP + r4) * XROW + o1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        m0[e] = fmaxf(a0[e], b0[e]);
                        m1[e] = fmaxf(a1[e], b1[e]);
                    }
                } else {
                    m0 = *(const f32x4_t*)(X + (2 * NRP - 1) * XROW + o0);
                    m1 = *(const f32x4_t*)(X + (2 * NRP - 1) * XROW + o1);
                }
                // bias and ReLU commute with the max (both monotone): applied here, to the 8 pooled values of this thread, instead
                // of to the 32 conv outputs of every lane
                f32x4_t bb0 = {0.f, 0.f, 0.f, 0.f}, bb1 = {0.f, 0.f, 0.f, 0.f};
                if (!XPAIR) {
                    bb0 = *(const f32x4_t*)(lbias + c8 * 8);
                    bb1 = *(const f32x4_t*)(lbias + c8 * 8 + 4);
                }
                u32x4_t oh, ol;
                float mv[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mv[e] = fmaxf(XPAIR ? m0[e] : m0[e] + bb0[e], 0.f);
                    mv[4 + e] = fmaxf(XPAIR ? m1[e] : m1[e] + bb1[e], 0.f);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t hh, ll;
                    split2u(mv[2 * e], mv[2 * e + 1], hh, ll);
                    oh[e] = hh;
                    ol[e] = ll;
                }
                const size_t o = ((size_t)(prev.b * a.PH + ph) * a.PW + pw) * 64 + c8 * 8;
                gstore16(a.yh + o, oh);
                gstore16(a.yl + o, ol);
                ovf.see(oh);
            }
        }
        if (!cur.valid) break;

        // ---- multiply tile n: conv rows c + 2 rp, + 1, channel tile ci ------------------------------------------------------
        const char* pbuf = smem + pb * PATCH;
        const int ox = 2 * cur.pw0 - 1 + lrow;
        const int oy = cur.c + 2 * rp;
        // a tile is an EDGE tile when one of its conv outputs has a 7x7 window that leaves the image (the border-class
        // correction applies) or lies outside the conv map itself (masked out of the max): 13 % of the tiles at 1024^2
        const bool edge = cur.c <= 1 || 2 * (cur.c + TH - 1) + 4 > a.H || cur.pw0 == 0 || 2 * (2 * cur.pw0 + TW - 2) + 4 > a.W;
        f32x16_t acc[2];
        if (XPAIR) {   // the generic pair form sums FROM the bias, like the kernels it replaces (same fp32 roundings)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t b4 = *(const f32x4_t*)(lbias + ci * 32 + 8 * g + 4 * lhi);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][4 * g + e] = b4[e];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;   // (folds into the first MFMA's srcC = 0: no per-tile moves)
        }
        // 20 fragments = patch rows 0..4 of this wave's row pair x 4 s2d columns: patch row q feeds conv row j = 0 through filter
        // row R = q and conv row j = 1 through R = q - 1, so the two rows share 12 of their 16 + 16 fragments.  Fragment of step
        // s + 1 requested before the MFMAs of step s; the pinned read : MFMA interleave keeps hipcc from hoisting every read
        // (80 VGPRs) above the chain.
        const char* xrow = pbuf + lhi * PLANE + ((rp * 2) * QW + lrow) * 16;
        frag_t xf[2], xg[2];      // xg: the lo-plane fragments (XPAIR)
        xf[0] = *(const frag_t*)xrow;
        if (XPAIR) xg[0] = *(const frag_t*)(xrow + 2 * PLANE);
#pragma unroll
        for (int s = 0; s < 20; ++s) {
            const int q = s >> 2, ks = s & 3;
            if (s + 1 < 20) {
                xf[(s + 1) & 1] = *(const frag_t*)(xrow + (((s + 1) >> 2) * QW + ((s + 1) & 3)) * 16);
                if (XPAIR) xg[(s + 1) & 1] = *(const frag_t*)(xrow + 2 * PLANE + (((s + 1) >> 2) * QW + ((s + 1) & 3)) * 16);
            }
            if (!(DIR_STEMU8_ABL & 1)) {
                if (q < 4) {
                    acc[0] = DT::mfma32(wfh[q][ks], xf[s & 1], acc[0]);
                    if (XPAIR) acc[0] = DT::mfma32(wfh[q][ks], xg[s & 1], acc[0]);
                    acc[0] = DT::mfma32(wfl[q][ks], xf[s & 1], acc[0]);
                }
                if (q > 0) {
                    acc[1] = DT::mfma32(wfh[q - 1][ks], xf[s & 1], acc[1]);
                    if (XPAIR) acc[1] = DT::mfma32(wfh[q - 1][ks], xg[s & 1], acc[1]);
                    acc[1] = DT::mfma32(wfl[q - 1][ks], xf[s & 1], acc[1]);
                }
            }
            if (XPAIR) continue;      // (the pinned interleave below is the two-product form's)
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          // 1 DS read
            if (q > 0 && q < 4)
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                      // 4 MFMAs (both conv rows)
            else
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                      // 2 MFMAs (patch rows 0 and 4: one conv row)
        }

        // ---- conv outputs outside the conv map -> -inf (edge tiles only; every pool window holds a real output), 3-max
        // along the row in registers: valid in the ODD lanes, window (l - 1, l, l + 1) ---------------------------------
        if (edge) {   // + the border-class correction of this conv pixel (the folded bias itself is added at the emit)
            const bool col_in = (unsigned)ox < (unsigned)a.OW;
            const int cc = XPAIR ? 0 : border_class(min(max(ox, 0), a.OW - 1), a.W);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool in = col_in && (unsigned)(oy + j) < (unsigned)a.OH;
                const int rc = XPAIR ? 0 : border_class(min(max(oy + j, 0), a.OH - 1), a.H);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4_t c4 = {0.f, 0.f, 0.f, 0.f};
                    if (!XPAIR) c4 = *(const f32x4_t*)(lcorr + (rc * 6 + cc) * 64 + ci * 32 + 8 * g + 4 * lhi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = acc[j][4 * g + e];
                        acc[j][4 * g + e] = in ? t + c4[e] : -__builtin_inff();
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = acc[j][e];
                acc[j][e] = fmaxf(fmaxf(lane_left(v), v), lane_right(v));
            }
        // acc[0] = H (row 2 rp, pooled along the row), m = max(H, row 2 rp + 1) = M
        char* X = smem + X_OFF + xb * XBUF;
        const int px = lrow >> 1;
        const bool odd = (lrow & 1) != 0;
        if (rp == 0 && cur.t > 0) {   // finish the pooled row the previous tile's last row pair began: KP = max(M[3] of tile n - 1, H[0])
            const char* Xp = smem + X_OFF + (xb ^ 1) * XBUF;
            if (odd) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int off = px * 256 + (((8 * ci + 2 * g + lhi) ^ (px & 7)) << 4);
                    const f32x4_t m3 = *(const f32x4_t*)(Xp + (NRP - 1) * XROW + off);
                    f32x4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(m3[e], acc[0][4 * g + e]);
                    *(f32x4_t*)(X + (2 * NRP - 1) * XROW + off) = v;
                }
            }
        }
        if (odd) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int off = px * 256 + (((8 * ci + 2 * g + lhi) ^ (px & 7)) << 4);
                f32x4_t h, m;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = acc[0][4 * g + e];
                    m[e] = fmaxf(acc[0][4 * g + e], acc[1][4 * g + e]);
                }
                *(f32x4_t*)(X + rp * XROW + off) = m;
                if (rp > 0) *(f32x4_t*)(X + (NRP - 1 + rp) * XROW + off) = h;
            }
        }

        prev = cur;
        cur = nxt;
        nxt = pre;
        advance(pre);
        pb = pb == NPB - 1 ? 0 : pb + 1;
        xb ^= 1;
    }
    ovf.flush(a.ovf);
}

// ---- uint8 NHWC image -> space-to-depth plane of u / 256 -----------------------------------------------------------------
// out[b][y2][x2][(dy * 2 + dx) * 3 + c] = fp16(u(2 y2 + dy, 2 x2 + dx, c) / 256) (exact), channels 12..15 and pixels beyond an
// odd H / W = 0.  One thread per s2d pixel: 2 x 6 image bytes in, 32 bytes out.
__global__ void __launch_bounds__(256) prep_input_u8_kernel(const uint8_t* __restrict__ img, uint16_t* __restrict__ out, int B, int H,
                                                            int W, int H2, int W2) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * H2 * W2;
    if (idx >= total) return;
    const int x2 = (int)(idx % W2);
    const int y2 = (int)((idx / W2) % H2);
    const int b = (int)(idx / ((long)W2 * H2));
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int y = 2 * y2 + dy;
        if (y >= H) continue;
        const uint8_t* row = img + (((size_t)b * H + y) * W + 2 * x2) * 3;
        const int n = (2 * x2 + 1 < W) ? 6 : 3;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < n) v[dy * 6 + k] = (float)row[k] * 0.00390625f;
    }
    u32x4_t o0, o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o0[e] = FP16::pack(v[2 * e], v[2 * e + 1]);
        o1[e] = FP16::pack(v[8 + 2 * e], v[8 + 2 * e + 1]);
    }
    gstore16(out + idx * 16, o0);
    gstore16(out + idx * 16 + 8, o1);
}

// ---- host: fold ToTensor + Normalize + BatchNorm into the filter pair, the bias and the border table ----------------------
// w: conv1.weight [64][3][7][7]; scale / bias: bn1 folded (y = conv * scale + bias); mean / std: the preprocess constants.
// hi / lo: [64][4][4][16] in the s2d packing of engine.hip (tap r = 2 R + dy - 1, s = 2 S + dx - 1, channel (dy * 2 + dx) * 3 + c)
// of w' = w . scale . 256 / (255 std_c); b2 = bias - sum_{147 taps} w . scale . mean_c / std_c; corr[rc][cc][o] = the part of that
// sum whose taps fall OUTSIDE the image for border class (rc, cc) (border_class above).  Sums in double.
int fold_stem_u8(const float* w, const float* scale, const float* bias, const float* mean3, const float* std3,
                 std::vector<uint16_t>& hi, std::vector<uint16_t>& lo, std::vector<float>& b2, std::vector<float>& corr) {
    for (int c = 0; c < 3; ++c)
        if (!(std3[c] > 0.f)) return fail(DIR_ERR_INVALID, "stem_u8: preprocess std must be positive");
    const int lo_r[6] = {0, 3, 1, 0, 0, 0}, hi_r[6] = {6, 6, 6, 5, 4, 3};
    std::vector<float> packed((size_t)64 * 256, 0.f);
    b2.assign(64, 0.f);
    corr.assign((size_t)36 * 64, 0.f);
    for (int o = 0; o < 64; ++o) {
        double all = 0.0;
        double in[6][6];
        for (int rc = 0; rc < 6; ++rc)
            for (int cc = 0; cc < 6; ++cc) in[rc][cc] = 0.0;
        for (int c = 0; c < 3; ++c) {
            const double ms = (double)mean3[c] / (double)std3[c];
            for (int r = 0; r < 7; ++r)
                for (int s = 0; s < 7; ++s) {
                    const double ws = (double)w[(((size_t)o * 3 + c) * 7 + r) * 7 + s] * (double)scale[o];
                    all += ws * ms;
                    for (int rc = 0; rc < 6; ++rc)
                        for (int cc = 0; cc < 6; ++cc)
                            if (r >= lo_r[rc] && r <= hi_r[rc] && s >= lo_r[cc] && s <= hi_r[cc]) in[rc][cc] += ws * ms;
                    const int R = (r + 1) >> 1, dy = (r + 1) & 1, S = (s + 1) >> 1, dx = (s + 1) & 1;   // r = 2 R + dy - 1
                    packed[(((size_t)o * 4 + R) * 4 + S) * 16 + (dy * 2 + dx) * 3 + c] = (float)(ws * 256.0 / (255.0 * (double)std3[c]));
                }
        }
        b2[o] = (float)((double)bias[o] - all);
        for (int rc = 0; rc < 6; ++rc)
            for (int cc = 0; cc < 6; ++cc) corr[((size_t)rc * 6 + cc) * 64 + o] = (float)(all - in[rc][cc]);
    }
    hi.resize(packed.size());
    lo.resize(packed.size());
    for (size_t i = 0; i < packed.size(); ++i) {
        hi[i] = f32_to_f16_bits(packed[i]);
        if ((hi[i] & 0x7c00u) == 0x7c00u && std::isfinite(packed[i]))
            return fail(DIR_ERR_RANGE, "finalize: a BatchNorm-folded weight of conv1, folded with the image normalisation (" + std::to_string(packed[i]) +
                                           ") exceeds the fp16 range; use DIR_BF16 or DIR_F32");
        lo[i] = f32_to_f16_bits(packed[i] - f16_bits_to_f32(hi[i]));
    }
    return DIR_OK;
}

int prep_input_u8(const void* img, void* out, int B, int H, int W, hipStream_t stream) {
    if (!img || !out) return fail(DIR_ERR_INVALID, "prep_input_u8: null pointer");
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const long total = (long)B * H2 * W2;
    hipLaunchKernelGGL(prep_input_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const uint8_t*)img,
                       (uint16_t*)out, B, H, W, H2, W2);
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

// The RAW form (no prep_input_u8 launch, no space-to-depth plane) applies when the image rows keep 6-byte groups 2-byte aligned
// and the 32-bit buffer offsets reach every byte; DIRTORCH_AMD_STEM_U8_PREP / _WG8 keep the two-kernel form for A/B.
bool stem_pool_u8_raw_ok(const void* img, int B, int H, int W) {
    return img && (W % 2) == 0 && ((uintptr_t)img & 1) == 0 && (size_t)B * H * W * 3 < (1ull << 31) && !env().stem_u8_prep &&
           !env().stem_u8_wg8;
}

int stem_pool_u8_launch(const void* img, const void* s2d, const void* w_hi, const void* w_lo, const float* bias, const float* corr,
                        void* y_hi, void* y_lo, int B, int H, int W, hipStream_t stream, int* ovf, int seg_tiles) {
    const bool raw = stem_pool_u8_raw_ok(img, B, H, W);
    if ((!raw && !s2d) || !w_hi || !w_lo || !bias || !corr || !y_hi || !y_lo) return fail(DIR_ERR_INVALID, "stem_pool_u8: null pointer");
    if (H < 7 || W < 7) return fail(DIR_ERR_INVALID, "stem_pool_u8: image smaller than the 7x7 stem");
    StemU8Args a;
    a.img = (const uint8_t*)img;
    a.img_bytes = raw ? (uint32_t)((size_t)B * H * W * 3) : 0;
    a.x = (const uint16_t*)s2d;
    a.xl = nullptr;
    a.wh = (const uint16_t*)w_hi;
    a.wl = (const uint16_t*)w_lo;
    a.bias = bias;
    a.corr = corr;
    a.yh = (uint16_t*)y_hi;
    a.yl = (uint16_t*)y_lo;
    a.B = B; a.H = H; a.W = W;
    a.H2 = (H + 1) / 2;
    a.W2 = (W + 1) / 2;
    a.OH = (H - 1) / 2 + 1;     // floor((H + 6 - 7) / 2) + 1
    a.OW = (W - 1) / 2 + 1;
    a.PH = (a.OH - 1) / 2 + 1;
    a.PW = (a.OW - 1) / 2 + 1;
    if (!raw && (size_t)B * a.H2 * a.W2 * 32 >= (1ull << 31))
        return fail(DIR_ERR_INVALID, "stem_pool_u8: input exceeds 2^31 bytes; lower the batch");
    a.x_bytes = raw ? 0 : (uint32_t)((size_t)B * a.H2 * a.W2 * 32);
    a.tiles_x = (a.PW + 14) / 15;
    // NRP = 2 (two 4-wave workgroups per CU) unless DIRTORCH_AMD_STEM_U8_WG8 asks for the one-workgroup form.
    // Segment length: seg_rows = 2 T' - 1 pooled rows are exactly T' tiles of TH conv rows (no wasted row); 31 unless that leaves
    // workgroup slots without an item.  seg_tiles (DIRTORCH_AMD_STEM_U8_SEG) counts tiles of EIGHT conv rows: 1 = 3 pooled rows.
    const bool wg8 = env().stem_u8_wg8;
    const int cus = cu_count();
    const int slots = wg8 ? cus : 2 * cus;
    int T = seg_tiles > 0 ? seg_tiles : 8;
    while (seg_tiles <= 0 && T > 1 && (long)B * a.tiles_x * ((a.PH + 4 * T - 2) / (4 * T - 1)) < 2L * slots) T >>= 1;
    a.seg_rows = 4 * T - 1;
    a.nseg = (a.PH + a.seg_rows - 1) / a.seg_rows;
    a.nitems = B * a.tiles_x * a.nseg;
    a.ovf = ovf;
    const int grid = a.nitems < slots ? a.nitems : slots;
    if (wg8) {
        constexpr int LDS = 3 * 16384 + 2 * 8 * 4096 + 256 + 36 * 64 * 4;
        static std::atomic<uint64_t> attr{0};
        DIR_HIP_CHECK(ensure_dynamic_lds((const void*)(stem_pool_u8_kernel<4, false>), LDS, attr));
        hipLaunchKernelGGL((stem_pool_u8_kernel<4, false>), dim3((unsigned)grid), dim3(512), LDS, stream, a);
    } else {
        constexpr int LDS = 3 * 8192 + 2 * 4 * 4096 + 256 + 36 * 64 * 4;
        static std::atomic<uint64_t> attr{0};
        static std::atomic<uint64_t> attr_raw{0};
        DIR_HIP_CHECK(ensure_dynamic_lds((const void*)(stem_pool_u8_kernel<2, false>), LDS, attr));
        DIR_HIP_CHECK(ensure_dynamic_lds((const void*)(stem_pool_u8_kernel<2, true>), LDS, attr_raw));
        static const bool dbg = getenv("DIRTORCH_AMD_DEBUG_OCCUPANCY") != nullptr;   // (read once; a debugging aid, not an A/B switch)
        static std::atomic<int> told{0};
        if (dbg && !told.exchange(1)) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)(stem_pool_u8_kernel<2, true>), 256, LDS);
            fprintf(stderr, "stem_pool_u8_kernel<2>: %d workgroups of 256 threads per CU at %d bytes of LDS (grid %d, %d items)\n", nb, LDS, grid,
                    a.nitems);
        }
        if (raw)
            hipLaunchKernelGGL((stem_pool_u8_kernel<2, true>), dim3((unsigned)grid), dim3(256), LDS, stream, a);
        else
            hipLaunchKernelGGL((stem_pool_u8_kernel<2, false>), dim3((unsigned)grid), dim3(256), LDS, stream, a);
    }
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

// ---- the generic paired stem on this kernel (XPAIR): any fp32 image as an fp16 pair, nothing folded -------------------------
// s2d_hi / s2d_lo: prep_input_pair's planes [B, H2, W2, 16]; w_hi / w_lo: the BatchNorm-folded filter pair [64][4][4][16]; bias:
// bn1's folded bias.  Same sums, in the same order, as conv_pair.hip's stem_pool_pair_persist_kernel (bit-identical outputs).
// img_f32 (optional): the fp32 NCHW image [B, 3, H, W] itself - with an even W (and 8-byte aligned, < 2^31 bytes) the kernel splits
// it into the pair on its own and s2d_hi / s2d_lo are not read (stem_pool_pair_raw_ok); H, W: the image size (only used then).
bool stem_pool_pair_raw_ok(const void* img_f32, int B, int H, int W) {
    return img_f32 && (W % 2) == 0 && ((uintptr_t)img_f32 & 7) == 0 && (size_t)B * 3 * H * W * 4 < (1ull << 31) && !env().stem_u8_prep &&
           !env().stem_pair_old && !env().stem_v1;
}

int stem_pool_pair_walk_launch(const void* s2d_hi, const void* s2d_lo, const void* w_hi, const void* w_lo, const float* bias,
                               void* y_hi, void* y_lo, int B, int H2, int W2, int OH, int OW, hipStream_t stream, int* ovf,
                               const void* img_f32, int H, int W) {
    const bool raw = stem_pool_pair_raw_ok(img_f32, B, H, W);
    if ((!raw && (!s2d_hi || !s2d_lo)) || !w_hi || !w_lo || !bias || !y_hi || !y_lo) return fail(DIR_ERR_INVALID, "stem_pool_pair: null pointer");
    if (!raw && (size_t)B * H2 * W2 * 32 >= (1ull << 31)) return fail(DIR_ERR_INVALID, "stem_pool_pair: input exceeds 2^31 bytes; lower the batch");
    StemU8Args a;
    a.img = (const uint8_t*)(raw ? img_f32 : nullptr);
    a.img_bytes = raw ? (uint32_t)((size_t)B * 3 * H * W * 4) : 0;
    a.x = (const uint16_t*)s2d_hi;
    a.xl = (const uint16_t*)s2d_lo;
    a.wh = (const uint16_t*)w_hi;
    a.wl = (const uint16_t*)w_lo;
    a.bias = bias;
    a.corr = nullptr;
    a.yh = (uint16_t*)y_hi;
    a.yl = (uint16_t*)y_lo;
    a.B = B;
    a.H = raw ? H : 2 * OH - 1;   // (the DMA form only uses them to decide which tiles take the masked path: the smaller image of this
    a.W = raw ? W : 2 * OW - 1;   // conv size is the safe one; the raw form addresses the image with them)
    a.H2 = H2; a.W2 = W2; a.OH = OH; a.OW = OW;
    a.PH = (OH - 1) / 2 + 1;
    a.PW = (OW - 1) / 2 + 1;
    a.x_bytes = raw ? 0 : (uint32_t)((size_t)B * H2 * W2 * 32);
    a.tiles_x = (a.PW + 14) / 15;
    const int cus = cu_count(), slots = 2 * cus;
    int T = 8;
    while (T > 1 && (long)B * a.tiles_x * ((a.PH + 4 * T - 2) / (4 * T - 1)) < 2L * slots) T >>= 1;
    a.seg_rows = 4 * T - 1;
    a.nseg = (a.PH + a.seg_rows - 1) / a.seg_rows;
    a.nitems = B * a.tiles_x * a.nseg;
    a.ovf = ovf;
    constexpr int LDS = 2 * 16384 + 2 * 4 * 4096 + 256 + 36 * 64 * 4;
    static std::atomic<uint64_t> attr{0};
    static std::atomic<uint64_t> attr_raw{0};
    DIR_HIP_CHECK(ensure_dynamic_lds((const void*)(stem_pool_u8_kernel<2, false, true>), LDS, attr));
    DIR_HIP_CHECK(ensure_dynamic_lds((const void*)(stem_pool_u8_kernel<2, true, true>), LDS, attr_raw));
    const int grid = a.nitems < slots ? a.nitems : slots;
    if (raw)
        hipLaunchKernelGGL((stem_pool_u8_kernel<2, true, true>), dim3((unsigned)grid), dim3(256), LDS, stream, a);
    else
        hipLaunchKernelGGL((stem_pool_u8_kernel<2, false, true>), dim3((unsigned)grid), dim3(256), LDS, stream, a);
    DIR_HIP_CHECK(hipGetLastError());
    return DIR_OK;
}

}  // namespace dir
