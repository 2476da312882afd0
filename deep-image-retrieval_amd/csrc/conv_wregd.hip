// conv_wregd.hip — the two-source 1x1 GEMM of layer2's first block with the weights stationary in REGISTERS (gfx950).
//
// relu([W3 | Wds] . [t2 ; x_s] + b3 + bds): conv3 + bn3 + the 1x1 stride-2 downsample branch + add + ReLU of the first
// block of layer2 (dirtorch/nets/backbones/resnet.py:78-85 with :134-141) as ONE GEMM whose K runs over two tensors,
// K = 128 (t2) + 256 (the block input, every second pixel of every second row) -> 512 channels.  On the persistent
// 256 x 256 tile (conv_persist.hip DUAL) this launch is six K-steps per tile between a fill and a 128 KB epilogue and
// re-fetches its 196 KB weight slice from L2 for every 256 pixels: 0.30 ms at batch 32 = 0.39 of the HBM roof.  K is
// short, so - as in conv_wreg.hip - a wave KEEPS its weights: 32 output channels x 384 inputs are 24 MFMA A-fragments
// = 96 VGPRs; a persistent 8-wave workgroup owns 256 consecutive output channels and streams 64-pixel tiles of the two
// sources (16 + 32 KB, double-buffered, staged through registers), shared by all eight waves.  The pixel operand is
// read Cout / 256 = 2 times (the second time from the XCD's L2: the two channel slices of a pixel tile are neighbours
// in the XCD-aware order), the weights once per workgroup.
//
// Epilogue: accumulators start at the bias (conv_persist.hip's convention: same MFMA, same K order - the sums are those
// of the DUAL ring kernel bit for bit); every lane applies ReLU, packs its 16 channels of one pixel and writes them to a
// [64 px][256 ch] 16-bit staging tile; after one barrier the whole workgroup stores the tile as 512-byte pixel rows
// (16 bytes per lane, consecutive lanes consecutive chunks).
#include "dir_common.h"
#include "conv_igemm.h"

namespace dir {

static constexpr uint32_t kOOBd = 0x80000000u;

__device__ __forceinline__ uint32_t fast_div_d(uint32_t n, uint32_t mul, uint32_t shr) {
    return mul ? (__umulhi(n, mul) >> shr) : n;
}

// KB1 / KB2 = 64-channel blocks of the first (flat) / second (strided) source
template <class DT, int KB1, int KB2>
__global__ void __launch_bounds__(512) conv1x1_wregd_kernel(const ConvArgs a) {
    constexpr int BM = 64;                     // pixels per step
    constexpr int BNW = 32;                    // channels per wave
    constexpr int BNG = 8 * BNW;               // channels per workgroup
    constexpr int KB = KB1 + KB2;
    constexpr int KS = KB * 4;                 // 16-wide k-slices
    constexpr int XBUF = KB * BM * 128;        // one input tile: KB blocks of [64 px][128 B]
    constexpr int SROW = BNG * 2 + 16;         // staging row: 256 packed channels + pad (bank spread, 16-byte aligned)
    constexpr int STG_OFF = 2 * XBUF;          // staging above the two input buffers
    typedef typename DT::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;

    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x2 =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2_bytes, 0x00020000);
    const uint32_t y_bytes = (uint32_t)((size_t)a.M * a.Cout * 2);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, y_bytes, 0x00020000);

    // work split: workgroup g serves channel slice g % nsl, pixel tiles (g / nsl) + i * (G / nsl); the nsl slices of
    // a pixel tile sit on the same XCD (conv_wreg.hip)
    const int nsl = a.Cout / BNG;
    const int mt = (a.M + BM - 1) / BM;
    const int per = gridDim.x / nsl;
    const int lid = a.no_xcd_map ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int sl = lid % nsl;
    int tile = lid / nsl;
    if (tile >= mt) return;
    Ovf<DT> ovf;
    const int n_wave = sl * BNG + wave * BNW;  // first output channel of this wave

    // ---- weights -> registers, once: A-fragment of k-slice ks (rows = this wave's 32 channels) --------------
    frag_t wf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        wf[ks] = *(const DIR_GLOBAL frag_t*)(a.w + (size_t)(n_wave + lrow) * a.Ktot + ks * 16 + 8 * lhi);
    // bias of the 16 channels a lane accumulates: rows 8 g + 4 lhi + e of the wave's channel tile
    f32x4_t bz[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bz[g] = *(const DIR_GLOBAL f32x4_t*)(a.bias + n_wave + 8 * g + 4 * lhi);
    // "already in registers" (conv_wreg.hip: keeps the wait for these loads out of the loop)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wf[ks]));
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bz[g]));

    // ---- per-lane constants ------------------------------------------------------------------------------------
    // input tile image in LDS: block kb (64 channels), pixel row p, 16-byte chunk c at
    //   kb*8192 + p*128 + ((c ^ ((p >> 1) & 7)) << 4)           (conv_igemm's swizzle); staged through registers
    const int spix = (tid >> 3) & 63;                  // pixel row this lane stages
    const int sslot = tid & 7;                         // 16-byte chunk of the 128-byte row
    const int sdst = spix * 128 + ((sslot ^ ((spix >> 1) & 7)) << 4);
    const int lswz = (lane >> 1) & 7;
    const int lbase = lrow * 128;

    // input tile t -> KB registers per lane (out-of-range rows read zeros)
    auto load_x = [&](int t, u32x4_t* xr) {
        const int m = t * BM + spix;
        const bool in = m < a.M;
        const uint32_t base = in ? (uint32_t)((m * a.Cin + sslot * 8) * 2) : kOOBd;
#pragma unroll
        for (int i = 0; i < KB1; ++i) xr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, base, i * 128, 0);
        // output pixel -> pixel (oh * stride2, ow * stride2) of the second source
        const uint32_t mm = in ? (uint32_t)m : 0u;
        const uint32_t b = fast_div_d(mm, a.div_ohw_mul, a.div_ohw_shr);
        const uint32_t rem = mm - b * (uint32_t)(a.OH * a.OW);
        const uint32_t oh = fast_div_d(rem, a.div_ow_mul, a.div_ow_shr);
        const uint32_t ow = rem - oh * (uint32_t)a.OW;
        const uint32_t base2 =
            in ? (uint32_t)((((b * a.H2 + oh * a.stride2) * a.W2 + ow * a.stride2) * a.Cin2 + sslot * 8) * 2) : kOOBd;
#pragma unroll
        for (int i = 0; i < KB2; ++i) xr[KB1 + i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x2, base2, i * 128, 0);
    };
    auto store_x = [&](const u32x4_t* xr, char* buf) {
#pragma unroll
        for (int i = 0; i < KB; ++i) *(u32x4_t*)(buf + i * (BM * 128) + sdst) = xr[i];
    };

    char* const stg = smem + STG_OFF;
    // this lane's staging writes: pixel j * 32 + lrow, channels wave * 32 + 8 g + 4 lhi .. + 3 (8 bytes)
    char* const swr = stg + lrow * SROW + (wave * BNW + 4 * lhi) * 2;
    // ... and its share of the tile's stores: 16-byte chunk q = tid + 512 k -> pixel q / 32, chunk q % 32
    const int cpix = tid >> 5, cchunk = tid & 31;
    const char* const srd = stg + cpix * SROW + cchunk * 16;
    const uint32_t ycol = (uint32_t)((sl * BNG) * 2 + cchunk * 16);

    u32x4_t xr[KB];
    load_x(tile, xr);
    int cur = 0;
    store_x(xr, smem);
    ring_barrier();   // first tile staged
    for (;;) {
        const bool more = tile + per < mt;
        const int next = more ? tile + per : tile;   // last step: a harmless repeat
        load_x(next, xr);                            // lands during this tile's MFMAs and stores
        const int m0 = tile * BM;

        const char* xb = smem + cur * XBUF;
        f32x16_t acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][4 * g + e] = bz[g][e];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const char* p = xb + (ks >> 2) * (BM * 128) + lbase + (((2 * (ks & 3) + lhi) ^ lswz) << 4);
            const frag_t x0 = *(const frag_t*)p;
            const frag_t x1 = *(const frag_t*)(p + 32 * 128);
            acc[0] = DT::mfma32(wf[ks], x0, acc[0]);
            acc[1] = DT::mfma32(wf[ks], x1, acc[1]);
            if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- ReLU, pack, stage: a lane holds 16 channels of pixel j * 32 + lrow ---------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4] = {acc[j][4 * g + 0], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                u32x2_t ov;
                ov[0] = DT::pack(v[0], v[1]);
                ov[1] = DT::pack(v[2], v[3]);
                ovf.see(ov);
                *(u32x2_t*)(swr + j * (32 * SROW) + g * 16) = ov;
            }
        ring_barrier();   // the staged tile is complete (and every wave is done with input buffer `cur`)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4_t ov = *(const u32x4_t*)(srd + k * (16 * SROW));
            const int m = m0 + k * 16 + cpix;
            const uint32_t off = m < a.M ? (uint32_t)m * (uint32_t)(a.Cout * 2) + ycol : kOOBd;
            __builtin_amdgcn_raw_buffer_store_b128(ov, rsrc_y, off, 0, 0);
        }
        if (!more) break;
        // publish the next tile into the buffer last read one step ago; the barrier also closes this step's
        // staging reads before the next step's staging writes
        store_x(xr, smem + (cur ^ 1) * XBUF);
        tile = next;
        cur ^= 1;
        ring_barrier();
    }
    ovf.flush(a.ovf);
}

bool conv1x1_wregd_admissible(const ConvArgs& a) {
    return a.x2 && a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.OH && a.W == a.OW && !a.res &&
           a.ksplit <= 1 && a.Cout % 256 == 0 && a.Cin == 128 && a.Cin2 == 256 && a.Ktot == a.Cin + a.Cin2;
}

template <class DT, int KB1, int KB2>
static hipError_t launch_wregd(const ConvArgs& a, hipStream_t stream) {
    constexpr int XBUF = (KB1 + KB2) * 64 * 128;
    constexpr int LDS = 2 * XBUF + 64 * (256 * 2 + 16);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv1x1_wregd_kernel<DT, KB1, KB2>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.x2_bytes = (uint32_t)((size_t)a.B * a.H2 * a.W2 * a.Cin2 * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    auto fd = [](uint32_t d, uint32_t& mul, uint32_t& shr) {   // exact n / d for n < 2^31 (conv_igemm.hip's constants)
        if (d <= 1) { mul = 0; shr = 0; return; }
        uint32_t l = 0;
        while ((1ull << l) < d) ++l;
        mul = (uint32_t)(((1ull << (31 + l)) + d - 1) / d);
        shr = l - 1;
    };
    fd((uint32_t)(a.OH * a.OW), b.div_ohw_mul, b.div_ohw_shr);
    fd((uint32_t)a.OW, b.div_ow_mul, b.div_ow_shr);
    const int nsl = a.Cout / 256;
    const int mt = (a.M + 63) / 64;
    int per = cu_count() / nsl;                // one persistent workgroup per CU
    if (per < 1) per = 1;
    if (per > mt) per = mt;
    b.no_xcd_map = env().no_xcdmap;
    hipLaunchKernelGGL(kern, dim3(per * nsl), dim3(512), LDS, stream, b);
    return hipGetLastError();
}

hipError_t conv1x1_wregd_bf16(const ConvArgs& a, hipStream_t stream) { return launch_wregd<BF16, 2, 4>(a, stream); }
hipError_t conv1x1_wregd_fp16(const ConvArgs& a, hipStream_t stream) { return launch_wregd<FP16, 2, 4>(a, stream); }

}  // namespace dir
