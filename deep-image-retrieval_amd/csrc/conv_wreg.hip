// conv_wreg.hip — 1x1 convolution with the weights stationary in REGISTERS (gfx950).
//
// The residual 1x1 convs that close the layer2/layer3 bottlenecks (K = 128 / 256 -> 512 / 1024
// channels, + residual + ReLU; dirtorch/nets/backbones/resnet.py:61-63,78-85) are bound by what a CU's
// memory pipe can move (~35 GB/s per CU measured, LDS-DMA fills + residual loads + stores together),
// and in the tiled kernels more than half of that is re-fetching the weight slice for every pixel
// tile (profiles/: 805 MB of L2->LDS fill per launch for 603 MB of HBM bytes on layer3).  K is short
// here, so a wave can simply KEEP its weights: 64 output channels x K = 256 inputs are 16 MFMA
// A-fragments x 2 channel tiles = 128 VGPRs.  A persistent 8-wave workgroup owns 512 consecutive
// output channels (64 per wave), loads them once, and then streams pixel tiles of 64: only the input
// tile (32 KB for K = 256, double-buffered, staged through registers) is filled per step and shared by all
// eight waves, so the input is re-read Cout/512 times instead of Cout/256 and the weights never.
//
// Everything else is the conv_igemm design: swapped MFMA roles (A = weights, B = pixels), XOR-swizzled
// 128-byte LDS rows (swizzle applied to the DMA source chunk), LDS-staged fp32 epilogue with 16-byte
// stores, K order = channel order.  The bias is added in the epilogue (the tiled kernels start their
// accumulators at it), so results agree with them to fp32 rounding, not bit for bit.
#include "dir_common.h"
#include "conv_igemm.h"

namespace dir {

static constexpr uint32_t kOOBr = 0x80000000u;

// KB = K / 64 (2 or 4): 64-channel blocks of the input
template <class DT, int KB>
__global__ void __launch_bounds__(512) conv1x1_wreg_kernel(const ConvArgs a) {
    constexpr int BM = 64;                     // pixels per step
    constexpr int BNW = 64;                    // channels per wave
    constexpr int NT = 512;
    constexpr int KS = KB * 4;                 // 16-wide k-slices
    constexpr int XBUF = KB * BM * 128;        // one input tile: KB blocks of [64 px][128 B]
    constexpr int NX = XBUF / 16 / NT;         // DMA instructions per lane per tile (KB = 4: 4)
    constexpr int EROW = 2 * 128 + 16;         // staging row: 64 fp32 + pad
    constexpr int EPI_OFF = 2 * XBUF;          // staging above the two input buffers
    constexpr int BIAS_OFF = EPI_OFF + 8 * 32 * EROW;   // 512 fp32 bias values after the staging rows
    typedef typename DT::frag_t frag_t;
    static_assert(NX >= 1 && XBUF % (16 * NT) == 0, "tile split");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;

    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    // residual loads and output stores go through bounds-checked descriptors too: a ragged last tile
    // needs no branch (out-of-range rows read zeros / drop the store), and with every VMEM op of the
    // loop unconditional the compiler's vmcnt bookkeeping is exact instead of "wait for everything"
    const uint32_t y_bytes = (uint32_t)((size_t)a.M * a.Cout * 2);
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, y_bytes, 0x00020000);

    // work split: workgroup g serves channel slice g % nsl, pixel tiles (g / nsl) + i * (G / nsl)
    const int nsl = a.Cout / 512;
    const int mt = (a.M + BM - 1) / BM;
    const int per = gridDim.x / nsl;           // workgroups per channel slice (launcher: G % nsl == 0)
    // the nsl channel slices of a pixel tile on the SAME XCD (consecutive ids of the remapped order), so that the
    // pixel operand comes from HBM once per tile instead of once per slice (layer3: PMC traffic 1.13x -> the 67 MB
    // of conv2's output were fetched by two L2s)
    const int lid = a.no_xcd_map ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int sl = lid % nsl;
    int tile = lid / nsl;
    if (tile >= mt) return;
    Ovf<DT> ovf;
    const int n_wave = sl * 512 + wave * BNW;  // first output channel of this wave

    // ---- weights -> registers, once: A-fragment of channel tile i, k-slice ks ------------------
    frag_t wf[2][KS];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wf[i][ks] = *(const DIR_GLOBAL frag_t*)(a.w + (size_t)(n_wave + i * 32 + lrow) * a.Ktot + ks * 16 +
                                                    8 * lhi);

    // Pin the weights as "already in registers": without this the compiler keeps its wait for these
    // loads at their first use INSIDE the loop, where a vmcnt(0) would also wait for the next tile's
    // DMA and the residual prefetch on every step.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wf[i][ks]));

    // ---- per-lane constants --------------------------------------------------------------------------
    // input tile image in LDS: block kb (64 channels), pixel row p, 16-byte chunk c at
    //   kb*8192 + p*128 + ((c ^ ((p >> 1) & 7)) << 4)           (conv_igemm's swizzle)
    // Staging goes through REGISTERS (buffer_load -> ds_write), not LDS-DMA: this loop keeps ordinary
    // loads (residual prefetch) in flight next to the input prefetch, and a counted vmcnt is only a
    // guarantee among loads that return in issue order - LDS-DMA and VGPR loads do not (measured: with
    // the input on LDS-DMA the first tile of every workgroup was computed from a half-landed buffer).
    // With one kind of load every wait is the compiler's own exact count.
    const int spix = (tid >> 3) & 63;                  // pixel row this lane stages (NT / 8 = 64 rows)
    const int sslot = tid & 7;                         // 16-byte chunk of the 128-byte row
    const int sdst = spix * 128 + ((sslot ^ ((spix >> 1) & 7)) << 4);
    const int lswz = (lane >> 1) & 7;
    const int lbase = lrow * 128;

    const int ecol = (lane & 7) * 8, erow = lane >> 3;   // epilogue: 8 lanes x 8 channels per pixel row
    // the workgroup's 512 bias values live in LDS (read back per epilogue pass: registers are all taken
    // by weights + accumulators + the prefetches)
    float* const sbias = (float*)(smem + BIAS_OFF);
    sbias[tid] = a.bias[sl * 512 + tid];
    const float* const bz = sbias + wave * BNW + ecol;

    // input tile t -> NX registers per lane (block i = channels 64 i ..; out-of-range rows read zeros)
    // (rev_m: the walk visits the pixel tiles from the last one - see ConvArgs::rev_m)
    auto phys = [&](int t) { return a.rev_m ? mt - 1 - t : t; };
    auto load_x = [&](int t, u32x4_t* xr) {
        const int m = phys(t) * BM + spix;
        const uint32_t base = m < a.M ? (uint32_t)((m * a.Cin + sslot * 8) * 2) : kOOBr;
#pragma unroll
        for (int i = 0; i < NX; ++i) xr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, base, i * 128, 0);
    };
    auto store_x = [&](const u32x4_t* xr, char* buf) {
#pragma unroll
        for (int i = 0; i < NX; ++i) *(u32x4_t*)(buf + i * (BM * 128) + sdst) = xr[i];
    };
    // residual of one 32-pixel strip (this wave's 64 channels): 4 x 16 B per lane
    const uint32_t ncol2 = (uint32_t)((n_wave + ecol) * 2);
    auto row_off = [&](int m) { return m < a.M ? (uint32_t)m * (uint32_t)(a.Cout * 2) + ncol2 : kOOBr; };
    auto load_res = [&](int t, int j, u32x4_t* r) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass)
            r[pass] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, row_off(phys(t) * BM + j * 32 + pass * 8 + erow), 0, 0);
    };

    u32x4_t xr[NX];
    load_x(tile, xr);
    int cur = 0;
    char* const ebase = smem + EPI_OFF + wave * (32 * EROW);
    // Residual prefetch, rolling: strip 0 of a tile is requested while the PREVIOUS tile computes its
    // strip 1, strip 1 at the top of the tile's own step - each about one tile's time ahead of its use,
    // enough for an HBM round trip under load, with 2 x 16 registers.
    u32x4_t rres0[4], rres1[4];
    load_res(tile, 0, rres0);
    store_x(xr, smem);
    ring_barrier();   // first tile staged (and the bias table written)
    for (;;) {
        const bool more = tile + per < mt;
        const int next = more ? tile + per : tile;   // last step: a harmless repeat
        load_x(next, xr);                            // lands during this tile's MFMAs
        const int m0 = phys(tile) * BM;
        load_res(tile, 1, rres1);

        const char* xb = smem + cur * XBUF;
        // the two 32-pixel strips of the tile one after the other: 32 accumulator registers instead of 64
        // (weights 128 + residual prefetch 32 + accumulators must stay under 256 without spilling - a
        // spill reload is a VMEM op and would drag a vmcnt(0) into the loop)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x16_t acc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const frag_t xf = *(const frag_t*)(xb + (ks >> 2) * (BM * 128) + j * (32 * 128) + lbase +
                                                   (((2 * (ks & 3) + lhi) ^ lswz) << 4));
                acc[0] = DT::mfma32(wf[0][ks], xf, acc[0]);
                acc[1] = DT::mfma32(wf[1][ks], xf, acc[1]);
                // bound how far the compiler hoists fragment reads: every register is spoken for
                if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue of the strip: acc -> LDS fp32 -> bias / residual / ReLU -> 16-byte stores ----
            // one wait for the strip's residual (requested long ago) instead of a counted wait per pass
            // that would also wait for this tile's own stores
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                if (j == 0) {
                    asm volatile("" : "+v"(rres0[pass]));
                } else {
                    asm volatile("" : "+v"(rres1[pass]));
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4_t v = {acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
                    *(f32x4_t*)(ebase + lrow * EROW + (i * 32 + 8 * g + 4 * lhi) * 4) = v;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int mrow = pass * 8 + erow;
                const int m = m0 + j * 32 + mrow;
                const f32x4_t f0 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4);
                const f32x4_t f1 = *(const f32x4_t*)(ebase + mrow * EROW + ecol * 4 + 16);
                {
                    const f32x4_t b0 = *(const f32x4_t*)bz, b1 = *(const f32x4_t*)(bz + 4);
                    float v[8] = {f0[0] + b0[0], f0[1] + b0[1], f0[2] + b0[2], f0[3] + b0[3],
                                  f1[0] + b1[0], f1[1] + b1[1], f1[2] + b1[2], f1[3] + b1[3]};
                    const u32x4_t rv = j == 0 ? rres0[pass] : rres1[pass];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo, hi;
                        DT::unpack(rv[e], lo, hi);
                        v[2 * e] += lo;
                        v[2 * e + 1] += hi;
                    }
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    u32x4_t ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = DT::pack(v[2 * e], v[2 * e + 1]);
                    __builtin_amdgcn_raw_buffer_store_b128(ov, rsrc_y, row_off(m), 0, 0);
                    ovf.see(ov);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (j == 0) load_res(next, 0, rres0);   // next tile's strip 0, one tile ahead
        }
        if (!more) break;
        // publish the next tile: its buffer was last read one step ago (every wave has passed that step's
        // barrier since), and this barrier also closes the reads of the buffer just used
        store_x(xr, smem + (cur ^ 1) * XBUF);
        tile = next;
        cur ^= 1;
        ring_barrier();
    }
    ovf.flush(a.ovf);
}

bool conv1x1_wreg_admissible(const ConvArgs& a) {
    // (the kernel is written for the residual convs: the residual prefetch is unconditional)
    return a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.OH && a.W == a.OW &&
           a.Cout % 512 == 0 && (a.Cin == 128 || a.Cin == 256) && a.res != nullptr;
}

template <class DT, int KB>
static hipError_t launch_wreg(const ConvArgs& a, hipStream_t stream) {
    constexpr int XBUF = KB * 64 * 128;
    constexpr int LDS = 2 * XBUF + 8 * 32 * (2 * 128 + 16) + 512 * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv1x1_wreg_kernel<DT, KB>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    const int nsl = a.Cout / 512;
    const int mt = (a.M + 63) / 64;
    int per = 256 / nsl;                       // one persistent workgroup per CU
    if (per > mt) per = mt;
    const bool no_xcd_map = env().no_xcdmap;   // A/B and bisecting
    b.no_xcd_map = no_xcd_map;
    hipLaunchKernelGGL(kern, dim3(per * nsl), dim3(512), LDS, stream, b);
    return hipGetLastError();
}

hipError_t conv1x1_wreg_launch(const ConvArgs& a, int dtype, hipStream_t stream) {
    if (a.Cin == 256)
        return dtype == DIR_BF16 ? launch_wreg<BF16, 4>(a, stream) : launch_wreg<FP16, 4>(a, stream);
    return dtype == DIR_BF16 ? launch_wreg<BF16, 2>(a, stream) : launch_wreg<FP16, 2>(a, stream);
}

}  // namespace dir
