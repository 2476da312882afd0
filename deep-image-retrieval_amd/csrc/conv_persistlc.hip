// conv_persistlc.hip — the persistent 256 x 256 1x1 GEMM with the work split by wave ROLE (gfx950).
//
// conv_persist.hip's deep-X form (three 32 KB slots for the pixel operand, two for the weights, one continuous K ring over
// all the tiles of a persistent workgroup) in conv_ring.hip's structure: waves 0-7 only read fragments, multiply and store
// their tile; waves 8-11 only issue LDS-DMA and wait for it.  In the one-role kernel every wave issues 8 LDS-DMA
// instructions per K-step between the barrier and its 32 MFMAs - 100-185 cycles each inside such a phase (the
// microarchitecture guide's price list), i.e. a third of the step on the layers whose matrix work is not hidden under
// memory time anyway: the wide 1x1 convs of layer4 and the two-source GEMMs of layers 3-4 (conv3 + downsample,
// dirtorch/nets/backbones/resnet.py:70-72, :78-85 with :134-141) run at 0.36-0.45 of the MFMA peak there, the 3x3 convs
// with the same tile and this role split (conv_patchw.hip) at 0.51-0.57.  conv_ring.hip had four consumer waves on a
// 128 x 256 tile and lost on exactly these layers ("four consumer waves - one per SIMD - multiply slower than eight");
// this is the eight-consumer form.
//
//   * tile 256 pixels x 256 channels, K-step 64; X(g + 2) and W(g + 1) are requested when the barrier of global step g
//     falls (g = tile index x T + k-step: the loaders run into the next tile while the consumers store this one);
//   * loaders: waves 8 / 9 the two halves of the pixel panel, 10 / 11 of the weight panel, 16 instructions of 1 KB per
//     stage each; a loader's queue holds nothing but its own LDS-DMA, so its counted vmcnt is exact;
//   * consumers: wave tile 64 x 128 (128 accumulator registers, 6 fragment reads per 8 MFMAs), two per SIMD;
//   * epilogue straight from the accumulators: + bias, ReLU, pack, v_permlane32_swap pairs the half-waves' 8-byte pieces
//     into 16-byte stores (conv_ring.hip).  All 160 KB of LDS belong to the ring, so the bias is ADDED here (the other 1x1
//     kernels start their accumulators at it): results agree with theirs to fp32 rounding, not bit for bit;
//   * DUAL: K-steps [0, Cin / 64) from x, the rest from x2 gathered at stride2 (conv_persist.hip's two-source form).
#include "dir_common.h"
#include "conv_igemm.h"

namespace dir {

static constexpr uint32_t kOOBp = 0x80000000u;

__device__ __forceinline__ void dma16p(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)lds, 16, voff, soff, 0, 0);
}

template <class DT, bool DUAL>
__global__ void __launch_bounds__(768) conv1x1_lc_kernel(const ConvArgs a) {
    constexpr int BM = 256, BN = 256;
    constexpr int TM = 2, TN = 4;              // consumer tile 64 pixels x 128 channels (4 x 2 consumer waves)
    constexpr int XS = BM * 128;               // 32 KB: one stage of either panel
    constexpr int WOFF = 3 * XS;               // X slots [0, 96K), W slots [96K, 160K)
    constexpr int NL = 16;                     // LDS-DMA instructions per loader wave and stage
    typedef typename DT::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];

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

    if (wave >= 8) {
        // ================================ loaders ======================================================================
        const int lw = wave - 8;
        const bool is_x = lw < 2;
        const int half = lw & 1;                           // rows [128 half, 128 half + 128) of the panel
        const int ahead = is_x ? 2 : 1;                    // stages this panel runs ahead of the consumers
        const int nslot = is_x ? 3 : 2;
        const int T1 = DUAL ? a.Cin / 64 : T;              // K-steps served by the first source
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)(is_x ? a.x : a.w), 0, is_x ? a.x_bytes : a.w_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc2 =
            __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? a.x2 : a.x), 0, DUAL ? a.x2_bytes : a.x_bytes, 0x00020000);
        const int dst0 = (is_x ? 0 : WOFF) + half * NL * 1024;
        // Instruction k of a stage covers rows 8 k .. 8 k + 7 of this wave's half (8 lanes x 16 B per 128-byte row); the chunks of a
        // row are XOR-swizzled with (row >> 1) & 7 on the SOURCE side (the LDS image is lane-linear).  For a flat source the
        // offsets of the 16 instructions are base + k x (8 rows), with the chunk term alternating by the parity of k - two
        // registers per tile instead of sixteen; only the strided second source keeps a table.
        const int rsub = lane >> 3;                                        // row of the lane inside an instruction
        const int chE = (lane & 7) ^ (lane >> 4), chO = (lane & 7) ^ (4 + (lane >> 4));
        const uint32_t rstride = (uint32_t)(8 * (is_x ? a.Cin : a.Ktot) * 2);   // bytes between instruction k and k + 1
        uint32_t baseE = 0, baseO = 0;
        uint32_t voff2[NL];   // (touched by the DUAL instantiations only)
        int m0 = 0;
        // (always_inline: left as a call, the lambda takes the by-value argument struct by reference and parks it in scratch)
        auto tile_offsets = [&](int tile) __attribute__((always_inline)) {
            const int tile_n = tile % a.tiles_n, tile_m = tile / a.tiles_n;
            if (is_x) {
                m0 = tile_m * BM + half * (NL * 8) + rsub;
                baseE = (uint32_t)((m0 * a.Cin + chE * 8) * 2);
                baseO = (uint32_t)((m0 * a.Cin + chO * 8) * 2);
                if constexpr (DUAL) {   // output pixel (b, oh, ow) -> pixel (b, oh * stride2, ow * stride2) of the second source
#pragma unroll
                    for (int k = 0; k < NL; ++k) {
                        const int m = m0 + 8 * k;
                        const uint32_t mm = m < a.M ? (uint32_t)m : 0u;
                        const uint32_t b = __umulhi(mm, a.div_ohw_mul) >> a.div_ohw_shr;
                        const uint32_t rem = mm - b * (uint32_t)(a.OH * a.OW);
                        const uint32_t oh = __umulhi(rem, a.div_ow_mul) >> a.div_ow_shr;
                        const uint32_t ow = rem - oh * (uint32_t)a.OW;
                        const uint32_t off2 = (((b * a.H2 + oh * a.stride2) * a.W2 + ow * a.stride2) * a.Cin2 + ((k & 1) ? chO : chE) * 8) * 2;
                        voff2[k] = m < a.M ? off2 : kOOBp;
                    }
                }
            } else {
                const int r0 = tile_n * BN + half * (NL * 8) + rsub;
                baseE = (uint32_t)((r0 * a.Ktot + chE * 8) * 2);
                baseO = (uint32_t)((r0 * a.Ktot + chO * 8) * 2);
                m0 = 0;
            }
        };
        int is_tile = first, is_t = 0, is_slot = 0;
        tile_offsets(first);
        const int mlim = is_x ? a.M : 0x7fffffff;
        auto issue_next = [&]() __attribute__((always_inline)) {
            char* dst = smem + is_slot * XS + dst0;
            bool second = false;
            if constexpr (DUAL) second = is_x && is_t >= T1;
            if (second) {
                if constexpr (DUAL) {
#pragma unroll
                    for (int k = 0; k < NL; ++k) dma16p(rsrc2, dst + k * 1024, voff2[k], (is_t - T1) * 128);
                }
            } else {
#pragma unroll
                for (int k2 = 0; k2 < NL / 2; ++k2) {
                    const uint32_t offe = m0 + 16 * k2 < mlim ? baseE + (uint32_t)(2 * k2) * rstride : kOOBp;
                    dma16p(rsrc, dst + (2 * k2) * 1024, offe, is_t * 128);
                    const uint32_t offo = m0 + 16 * k2 + 8 < mlim ? baseO + (uint32_t)(2 * k2 + 1) * rstride : kOOBp;
                    dma16p(rsrc, dst + (2 * k2 + 1) * 1024, offo, is_t * 128);
                }
            }
            is_slot = is_slot + 1 == nslot ? 0 : is_slot + 1;
            if (++is_t == T) {
                is_t = 0;
                is_tile += (int)gridDim.x;
                if (is_tile < ntiles) tile_offsets(is_tile);
            }
        };
        issue_next();
        if (is_x && total > 1) issue_next();
        for (int g = 0; g < total; ++g) {
            // this wave's part of stage g has landed; the pixel loaders leave their newest stage (g + 1) in flight
            if (is_x && g + 1 < total) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NL) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            ring_barrier();   // hand-off g: stage g is complete; the consumers have left step g - 1
            if (g + ahead < total) issue_next();
        }
        return;
    }

    // ==================================== consumers ===================================================================
    const int wn = wave & 1, wm = wave >> 1;
    const int lswz = (lane >> 1) & 7;
    int loff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) loff[ks] = lrow * 128 + (((2 * ks + lhi) ^ lswz) << 4);
    const int xfrag = (wm * TM * 32) * 128;
    const int wfrag = WOFF + (wn * TN * 32) * 128;
    Ovf<DT> ovf;
    f32x16_t acc[TN][TM];
    int tile = first, t = 0, xslot = 0, wslot = 0;
    for (int g = 0; g < total; ++g) {
        ring_barrier();   // hand-off g (see the loaders)
        if (t == 0) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
        const char* xs = smem + xslot * XS;
        const char* ws = smem + wslot * XS;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            frag_t wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = *(const frag_t*)(ws + wfrag + i * 4096 + loff[ks]);
#pragma unroll
            for (int j = 0; j < TM; ++j) xf[j] = *(const frag_t*)(xs + xfrag + j * 4096 + loff[ks]);
            // (the PIXEL fragment held over four consecutive MFMAs, the weight fragment changing: conv_patchw.hip's measured order
            // for the power-bound matrix pipe; per accumulator the k order is the other 1x1 kernels')
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int i = 0; i < TN; ++i) acc[i][j] = DT::mfma32(wf[i], xf[j], acc[i][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this step's LDS reads retired before the next hand-off
        if (t == T - 1) {
            // ---- tile complete: + bias, ReLU, pack; v_permlane32_swap on the pair (gg, gg + 1) leaves lanes 0-31 with
            //      channels 8 gg .. 8 gg + 7 and lanes 32-63 with the next eight: 16-byte stores (conv_ring.hip).
            //      acc[i][j][4 gg + e] = channel i*32 + 8 gg + 4 lhi + e of pixel j*32 + lrow.
            const int tile_n = tile % a.tiles_n, tile_m = tile / a.tiles_n;
            const int n_wave = tile_n * BN + wn * TN * 32;
            const int m_wave = tile_m * BM + wm * TM * 32;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                f32x4_t b4[4];
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) b4[gg] = *(const DIR_GLOBAL f32x4_t*)(a.bias + n_wave + i * 32 + 8 * gg + 4 * lhi);
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        uint32_t q2[2][2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float x = acc[i][j][4 * (2 * h + q) + e] + b4[2 * h + q][e];
                                v[e] = a.relu ? fmaxf(x, 0.f) : x;
                            }
                            q2[q][0] = DT::pack(v[0], v[1]);
                            q2[q][1] = DT::pack(v[2], v[3]);
                        }
                        const auto r0 = __builtin_amdgcn_permlane32_swap(q2[0][0], q2[1][0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(q2[0][1], q2[1][1], false, false);
                        const u32x4_t ov = {r0[0], r1[0], r0[1], r1[1]};
                        const int m = m_wave + j * 32 + lrow;
                        if (m < a.M) {
                            gstore16(a.y + ((size_t)m * a.Cout + n_wave + i * 32 + h * 16 + lhi * 8), ov);
                            ovf.see(ov);
                        }
                    }
            }
            tile += (int)gridDim.x;
            t = 0;
        } else {
            ++t;
        }
        xslot = xslot + 1 == 3 ? 0 : xslot + 1;
        wslot ^= 1;
    }
    ovf.flush(a.ovf);
}

bool conv1x1_lc_admissible(const ConvArgs& a) {
    const bool base = a.R == 1 && a.S == 1 && a.pad == 0 && a.stride == 1 && a.H == a.OH && a.W == a.OW && a.Cout % 256 == 0 &&
                      a.Cin % 64 == 0 && a.res == nullptr && a.ksplit <= 1;
    if (a.x2) return base && a.Cin2 % 64 == 0 && a.Ktot == a.Cin + a.Cin2 && a.OW > 1;
    return base && a.Cin >= 128;
}

template <class DT, bool DUAL>
static hipError_t launch_lc(const ConvArgs& a, hipStream_t stream) {
    constexpr int LDS = 5 * 256 * 128;   // three pixel slots + two weight slots: all 160 KiB
    static_assert(LDS <= 160 * 1024, "LDS map");
    auto kern = conv1x1_lc_kernel<DT, DUAL>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.T = a.Ktot / 64;
    b.tiles_m = ceil_div(a.M, 256);
    b.tiles_n = a.Cout / 256;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    if (DUAL) b.x2_bytes = (uint32_t)((size_t)a.B * a.H2 * a.W2 * a.Cin2 * 2);
    b.no_xcd_map = env().no_xcdmap;
    auto fd = [](uint32_t d, uint32_t& mul, uint32_t& shr) {   // exact n / d for 1 < d, n < 2^31 (conv_igemm.hip's constants)
        uint32_t l = 0;
        while ((1ull << l) < d) ++l;
        mul = (uint32_t)(((1ull << (31 + l)) + d - 1) / d);
        shr = l - 1;
    };
    if (DUAL) {
        fd((uint32_t)(a.OH * a.OW), b.div_ohw_mul, b.div_ohw_shr);
        fd((uint32_t)a.OW, b.div_ow_mul, b.div_ow_shr);
    }
    const int ntiles = b.tiles_m * b.tiles_n;
    const int ncu = cu_count();
    hipLaunchKernelGGL(kern, dim3(ntiles < ncu ? ntiles : ncu), dim3(768), LDS, stream, b);
    return hipGetLastError();
}

hipError_t conv1x1_lc_launch(const ConvArgs& a, int dtype, hipStream_t stream) {
    return dtype == DIR_BF16 ? launch_lc<BF16, false>(a, stream) : launch_lc<FP16, false>(a, stream);
}
hipError_t conv1x1_lc_dual_bf16(const ConvArgs& a, hipStream_t stream) { return launch_lc<BF16, true>(a, stream); }
hipError_t conv1x1_lc_dual_fp16(const ConvArgs& a, hipStream_t stream) { return launch_lc<FP16, true>(a, stream); }

}  // namespace dir
