// conv_ring.hip — persistent 1x1 convolution: loader waves feed a K ring that never drains, consumer waves multiply (gfx950).
//
// An EXPERIMENT that became a tuner candidate, not a default: the wide 1x1 convs without a residual (conv1 of the
// layer3 / layer4 bottlenecks, 512 ... 2048 -> 256 / 512 channels, dirtorch/nets/backbones/resnet.py:56,70-72) run at
// 0.49 of the HBM roof on conv_persist.hip, and this file was written to find out why.  What the probes and the
// phase-ablation builds said (profiles/r03_window_probe.txt, r03_ring_ablation*.txt, r03_read_store_mix_probe.txt):
//   * L2 hits ride nearly free next to HBM misses: a CU pulls 23.4 GB/s of HBM bytes (6.0 TB/s for the chip, the
//     read ceiling) AND the same rate of L2-resident bytes when ~100 KB of LDS-DMA requests are outstanding (one 64 KB
//     stage in flight: 20 + 20).  The "per-CU request window" of round 2 is not a byte budget shared by both kinds.
//   * When every wave does "issue the next stage, then multiply this one", the two phases do not overlap: a
//     memory-bound CU's request queue is full, so each LDS-DMA instruction holds its wave at issue until the queue
//     drains - all eight waves stall together, then multiply together (ring alone 46 us, MFMAs alone ~40 us, both
//     82 us for 1024 -> 256 at batch 32).  With the roles split - waves 4-6 only issue LDS-DMA and wait for it, waves
//     0-3 only read fragments and multiply (the producer / consumer structure of the CDNA guide's section 5.6) - the
//     ring plus the MFMAs take 48-50 us: fully overlapped, at the read ceiling.
//   * The remaining 30-40 us are the 67 MB of OUTPUT.  Every way of writing it cost the same: 16-byte stores straight
//     from the accumulators by the consumers, the same stores handed through an LDS mailbox to the loaders or to a
//     dedicated store wave that never waits on vmcnt, full 128-byte lines through an LDS staging area.  The probe
//     settles it: a read stream of 16 KB per step per CU drops from 6.2 to 4.1 TB/s the moment 4 KB per step are
//     WRITTEN next to it, whatever the store pattern (contiguous 4.08 + 1.02, 128-byte lines 4.14 + 1.03, 32-byte
//     pieces 3.8 + 0.95 TB/s) - HBM delivers ~5.1 TB/s of read + write traffic at any mix from 4:1 to 1:1 (the 1:1 copy
//     probe of round 2: 5.15).  The floor of this layer is therefore 335 MB / 5.1 TB/s = 66 us, not 54, and 80-90 us
//     is what four structurally different kernels all reach.
// Form kept here: the split-role ring with direct stores (the simplest of the four, 80 us standalone, within 2 us of
// conv_persist.hip inside the network; 9 us faster on the 512 -> 256 conv1 of layer3's first block).  It is in the
// autotuner's pool and selectable by name; the built-in heuristic keeps conv_persist.hip.
//   * tile 128 pixels x 256 channels, K-step 64: a stage is 16 KB of pixels + 32 KB of weights, THREE slots
//     (144 KB), two stages always in flight; pixels are read from HBM exactly once, the weight panel is re-streamed
//     from L2 once per 128 pixels;
//   * ONE ring over all the tiles of a persistent workgroup (global step g = tile index x T + k-step): the loaders
//     run into the next tile while the consumers finish this one - no fill at a tile boundary; one s_barrier per
//     K-step is the hand-off in both directions (stage g landed / slot g - 1 is free);
//   * consumer tile 64 pixels x 128 channels (128 accumulator registers, 6 fragment reads per 8 MFMAs, the reads of
//     K-slice ks + 1 issued before the MFMAs of ks: one wave per SIMD has nobody else to hide LDS latency behind);
//   * output straight from the accumulators: ReLU, pack, v_permlane32_swap pairs the two half-waves' 8-byte pieces
//     into 16-byte stores (CDNA guide T21) - no LDS staging, no extra barrier;
//   * bias in LDS; the loaders issue nothing but LDS-DMA, so their counted vmcnt sees one kind of op, in order.
// Everything else is the conv_igemm design: descriptor LDS-DMA with the K-step in the scalar offset, XOR-swizzled
// 128-byte rows, swapped MFMA roles (A = weights, B = pixels), accumulators that start at the bias; the K order and
// the fp32 sums are those of every other 1x1 kernel here (bit-identical outputs).
#include "dir_common.h"
#include "conv_igemm.h"

// Timing-only experiment builds (scripts/exp_abl.sh conv_ring DIR_RING_ABL <bits>): -DDIR_RING_ABL=<bits> compiles phases out - 1 = no pixel DMA,
// 2 = no weight DMA, 4 = no fragment reads / MFMAs, 8 = no epilogue.  Results are NOT valid convolutions.
#ifndef DIR_RING_ABL
#define DIR_RING_ABL 0
#endif

namespace dir {

static constexpr uint32_t kOOBr = 0x80000000u;

__device__ __forceinline__ void dma16r(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ uint32_t fast_div_r(uint32_t n, uint32_t mul, uint32_t shr) {
    return mul ? (__umulhi(n, mul) >> shr) : n;
}

// LDS map: [0, 144K) ring, [144K, 152K) bias (Cout <= 2048).
template <class DT>
__global__ void __launch_bounds__(512) conv1x1_ring_kernel(const ConvArgs a) {
    constexpr int BM = 128, BN = 256;
    constexpr int TM = 2, TN = 4;              // consumer tile 64 pixels x 128 channels (2 x 2 consumer waves)
    constexpr int XS = BM * 128, STAGE = (BM + BN) * 128;   // 16 KiB + 32 KiB
    constexpr int NSLOT = 3;
    constexpr int BIAS_OFF = NSLOT * STAGE;
    typedef typename DT::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lbias = (float*)(smem + BIAS_OFF);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;

    const int ntiles = a.tiles_m * a.tiles_n;
    const int T = a.T;
    const int first = a.no_xcd_map ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    if (first >= ntiles) return;
    const int my_tiles = (ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * T;

    for (int i = tid; i < a.Cout; i += 512) lbias[i] = a.bias[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the only VGPR-kind loads of the kernel
    ring_barrier();

    // Where piece p of the output tile `tile` goes, for the lane that holds (or mirrors) consumer wave cw's lane:
    // piece p = (j, i, h): pixel block j, channel block i, 16-channel half h.  Element offset into y, or -1 past M.
    const int yM = a.M, yC = a.Cout, ytn = a.tiles_n;
    auto piece_off = [=](int tile, int cw, int p) -> long {
        const int tile_n = tile % ytn, tile_m = tile / ytn;
        const int m = tile_m * BM + (cw >> 1) * TM * 32 + (p >> 3) * 32 + lrow;
        const int n = tile_n * BN + (cw & 1) * TN * 32 + ((p >> 1) & 3) * 32 + (p & 1) * 16 + lhi * 8;
        return m < yM ? (long)m * yC + n : -1L;
    };

    if (wave == 7) {
        // the eighth wave only keeps the barrier count (three loaders cover a stage in 16 instructions each)
        for (int g = 0; g < total; ++g) ring_barrier();
        return;
    }
    if (wave >= 4) {
        // ================================ loaders ==========================================================
        // A stage is 48 LDS-DMA instructions (1 KiB each): 16 for the pixel panel, 32 for the weight panel; three loader
        // waves take 16 each - wave 4 the pixels, waves 5 and 6 one half of the weights.  Instruction i of a panel covers
        // rows 8i .. 8i+7 (8 lanes x 16 B per row); the 16-byte chunks of a row are XOR-swizzled with (row >> 1) & 7 on
        // the SOURCE side (the LDS image is lane-linear).
        constexpr int NL = 16;
        const int lw = wave - 4;
        const bool is_x = lw == 0;
        const uint16_t* src_base = is_x ? a.x : a.w;
        const uint32_t src_bytes = is_x ? a.x_bytes : a.w_bytes;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src_base, 0, src_bytes, 0x00020000);
        const int dst0 = is_x ? 0 : XS + (lw - 1) * NL * 1024;
        uint32_t voff[NL];
        // (always_inline: left as a call, the lambda takes the by-value argument struct by reference and parks it in scratch)
        auto tile_offsets = [&](int tile) __attribute__((always_inline)) {
            const int tile_n = tile % a.tiles_n, tile_m = tile / a.tiles_n;
            if (is_x) {
#pragma unroll
                for (int k = 0; k < NL; ++k) {
                    const int row = k * 8 + (lane >> 3);
                    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                    const int m = tile_m * BM + row;
                    uint32_t off;
                    if (a.flat) {
                        off = (uint32_t)((m * a.Cin + chunk * 8) * 2);
                    } else {  // strided 1x1: output pixel (b, oh, ow) -> input pixel (b, oh * s, ow * s)
                        const uint32_t mm = m < a.M ? (uint32_t)m : 0u;
                        const uint32_t b = fast_div_r(mm, a.div_ohw_mul, a.div_ohw_shr);
                        const uint32_t rem = mm - b * (uint32_t)(a.OH * a.OW);
                        const uint32_t oh = fast_div_r(rem, a.div_ow_mul, a.div_ow_shr);
                        const uint32_t ow = rem - oh * (uint32_t)a.OW;
                        const uint32_t sh = (uint32_t)a.H, sw = (uint32_t)a.W, st = (uint32_t)a.stride, sc = (uint32_t)a.Cin;
                        off = (uint32_t)((((b * sh + oh * st) * sw + ow * st) * sc + chunk * 8) * 2);
                    }
                    voff[k] = m < a.M ? off : kOOBr;
                }
            } else {
#pragma unroll
                for (int k = 0; k < NL; ++k) {
                    const int row = ((lw - 1) * NL + k) * 8 + (lane >> 3);
                    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                    voff[k] = (uint32_t)(((tile_n * BN + row) * a.Ktot + chunk * 8) * 2);
                }
            }
        };
        int is_tile = first, is_t = 0, is_slot = 0;
        tile_offsets(first);
        auto issue_next = [&]() __attribute__((always_inline)) {
            char* dst = smem + is_slot * STAGE + dst0;
#pragma unroll
            for (int k = 0; k < NL; ++k)
                if (!(DIR_RING_ABL & (is_x ? 1 : 2))) dma16r(rsrc, dst + k * 1024, voff[k], is_t * 128);
            is_slot = is_slot + 1 == NSLOT ? 0 : is_slot + 1;
            if (++is_t == T) {
                is_t = 0;
                is_tile += (int)gridDim.x;
                if (is_tile < ntiles) tile_offsets(is_tile);
            }
        };
        issue_next();
        if (total > 1) issue_next();
        for (int g = 0; g < total; ++g) {
            // this wave's part of stage g has landed; its NL newest ops (stage g + 1) may stay in flight.  Nothing but
            // LDS-DMA ever enters this wave's queue, so the count is exact.
            if (g + 1 < total) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NL) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            ring_barrier();   // hand-off g: stage g is complete; the consumers have left step g - 1
            if (g + 2 < total) issue_next();
        }
        return;
    }

    // ==================================== consumers =========================================================
    const int wn = wave & 1, wm = wave >> 1;
    const int lswz = (lane >> 1) & 7;
    int loff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) loff[ks] = lrow * 128 + (((2 * ks + lhi) ^ lswz) << 4);
    const int xfrag = (wm * TM * 32) * 128;
    const int wfrag = XS + (wn * TN * 32) * 128;
    Ovf<DT> ovf;
    f32x16_t acc[TN][TM];
    int tile = first, t = 0, slot = 0;
    for (int g = 0; g < total; ++g) {
        ring_barrier();   // hand-off g (see the loaders)
        if (t == 0) {
            const int n_wave = (tile % a.tiles_n) * BN + wn * TN * 32;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const f32x4_t b4 = *(const f32x4_t*)(lbias + n_wave + i * 32 + 8 * gg + 4 * lhi);
#pragma unroll
                    for (int j = 0; j < TM; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * gg + e] = b4[e];
                }
        }
        const char* stage = smem + slot * STAGE;
        if (!(DIR_RING_ABL & 4)) {
            // the fragments of K-slice ks + 1 are requested before the MFMAs of ks: one wave per SIMD has nobody
            // else to hide its LDS latency behind
            frag_t wf[2][TN], xf[2][TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[0][i] = *(const frag_t*)(stage + wfrag + i * 4096 + loff[0]);
#pragma unroll
            for (int j = 0; j < TM; ++j) xf[0][j] = *(const frag_t*)(stage + xfrag + j * 4096 + loff[0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < TN; ++i)
                        wf[(ks + 1) & 1][i] = *(const frag_t*)(stage + wfrag + i * 4096 + loff[ks + 1]);
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        xf[(ks + 1) & 1][j] = *(const frag_t*)(stage + xfrag + j * 4096 + loff[ks + 1]);
                }
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc[i][j] = DT::mfma32(wf[ks & 1][i], xf[ks & 1][j], acc[i][j]);
            }
        }
        if (t == T - 1) {
            if (!(DIR_RING_ABL & 8)) {
                // ---- tile complete: ReLU, pack; v_permlane32_swap on the pair (gg, gg + 1) leaves lanes 0-31 with
                //      channels 8 gg .. 8 gg + 7 and lanes 32-63 with the next eight (CDNA guide T21): 16-byte stores.
                //      acc[i][j][4 gg + e] = channel i*32 + 8 gg + 4 lhi + e of pixel j*32 + lrow.
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            uint32_t q2[2][2];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                float v[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float x = acc[i][j][4 * (2 * h + q) + e];
                                    v[e] = a.relu ? fmaxf(x, 0.f) : x;
                                }
                                q2[q][0] = DT::pack(v[0], v[1]);
                                q2[q][1] = DT::pack(v[2], v[3]);
                            }
                            const auto r0 = __builtin_amdgcn_permlane32_swap(q2[0][0], q2[1][0], false, false);
                            const auto r1 = __builtin_amdgcn_permlane32_swap(q2[0][1], q2[1][1], false, false);
                            const u32x4_t ov = {r0[0], r1[0], r0[1], r1[1]};
                            const long o = piece_off(tile, wave, j * 8 + i * 2 + h);
                            if (o >= 0) {
                                gstore16(a.y + o, ov);
                                ovf.see(ov);
                            }
                        }
            }
            tile += (int)gridDim.x;
            t = 0;
        } else {
            ++t;
        }
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    ovf.flush(a.ovf);
}

bool conv1x1_ring_admissible(const ConvArgs& a) {
    return a.R == 1 && a.S == 1 && a.pad == 0 && a.Cout % 256 == 0 && a.Cout <= 2048 && a.Cin % 64 == 0 &&
           a.Cin >= 128 && a.res == nullptr && a.x2 == nullptr;
}

template <class DT>
static hipError_t launch_ring(const ConvArgs& a, hipStream_t stream) {
    constexpr int LDS = 3 * (128 + 256) * 128 + 2048 * 4;   // ring + bias
    static_assert(LDS <= 160 * 1024, "LDS map");
    auto kern = conv1x1_ring_kernel<DT>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.T = a.Ktot / 64;
    b.tiles_m = ceil_div(a.M, 128);
    b.tiles_n = a.Cout / 256;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    b.flat = (a.stride == 1 && a.H == a.OH && a.W == a.OW);
    const bool no_xcd_map = env().no_xcdmap;   // A/B and bisecting
    b.no_xcd_map = no_xcd_map;
    auto fd = [](uint32_t d, uint32_t& mul, uint32_t& shr) {   // exact n / d for n < 2^31 (as in conv_igemm.hip)
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

hipError_t conv1x1_ring_launch(const ConvArgs& a, int dtype, hipStream_t stream) {
    return dtype == DIR_BF16 ? launch_ring<BF16>(a, stream) : launch_ring<FP16>(a, stream);
}

}  // namespace dir
