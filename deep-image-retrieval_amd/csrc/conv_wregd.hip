// conv_wregd.hip — the two-source 1x1 GEMM of layer2's first block with the weights stationary in REGISTERS (gfx950).
//
// relu([W3 | Wds] . [t2 ; x_s] + b3 + bds): conv3 + bn3 + the 1x1 stride-2 downsample branch + add + ReLU of the first
// block of layer2 (dirtorch/nets/backbones/resnet.py:78-85 with :134-141) as ONE GEMM whose K runs over two tensors,
// K = 128 (t2) + 256 (the block input, every second pixel of every second row) -> 512 channels.  On the persistent
// 256 x 256 tile (conv_persist.hip DUAL) this launch is six K-steps per tile between a fill and a 128 KB epilogue and
// re-fetches its 196 KB weight slice from L2 for every 256 pixels: 0.33 ms at batch 32 = 0.35 of the HBM roof.  K is
// short, so - as in conv_wreg.hip - a wave KEEPS its weights: 32 output channels x 384 inputs are 24 MFMA A-fragments
// = 96 VGPRs; a persistent workgroup owns 256 consecutive output channels and streams 64-pixel tiles of the two
// sources (16 + 32 KB, double-buffered in LDS, two more tiles in flight in registers).  The pixel operand is read
// Cout / 256 = 2 times (the second time from the XCD's L2: the two channel slices of a pixel tile are neighbours in the
// XCD-aware order), the weights once per workgroup.  The work is split by wave ROLE (eight consumers, four memory
// waves: the kernel comment below); how the kernel got there, with the phase builds of every form, is
// profiles/r06_wregd.txt - 0.24 ms inside the network.
//
// Accumulators start at the bias (conv_persist.hip's convention: same MFMA, same K order - the sums are those of the DUAL
// ring kernel bit for bit); a consumer lane applies ReLU, packs its 16 channels of one pixel and writes them to a
// [64 px][256 ch] 16-bit staging tile (16-byte chunks XOR-swizzled by the pixel row); the memory waves store the tile as
// 512-byte pixel rows (16 bytes per lane, consecutive lanes consecutive chunks).
#include "dir_common.h"
#include "conv_igemm.h"

#ifndef DIR_WREGD_ABL   // experiment builds only (scripts/exp_abl.sh conv_wregd DIR_WREGD_ABL <bits>): 1 no MFMAs, 2 no stores,
#define DIR_WREGD_ABL 0 // 4 the second source read as a flat tensor (no stride), 8 no pixel loads - timing only
#endif

namespace dir {

static constexpr uint32_t kOOBd = 0x80000000u;

// KB1 / KB2 = 64-channel blocks of the first (flat) / second (strided) source
//
// TWELVE waves, two roles (round-3 finding, DESIGN.md section 3: a wave that issues memory instructions into a full queue
// is held at issue, and its MFMAs wait behind them - with every wave doing both, compute alone took 166 us, memory alone
// 189 us and the kernel 271 us, whether the memory work sat after the MFMAs or between them, profiles/r06_wregd.txt):
//   waves 0-7  CONSUMERS: 32 output channels x 384 inputs of weights in registers; per tile 48 MFMAs from the published
//              input tile, then ReLU / pack / stage.  They issue no global memory instruction inside the loop.
//   waves 8-11 MEMORY waves: request tile i + 2 (12 x 16 bytes per lane), store the staged tile i - 1 (8 x 16 bytes per
//              lane, 512-byte pixel rows), publish tile i + 1 into the other input buffer.  They never multiply.
// ONE barrier per tile joins the roles (tile i staged, tile i + 1 published, input buffer i % 2 and staging tile (i - 1) % 2
// free): the staging tile is double-buffered, so the memory waves never wait for the pack nor the consumers for the stores.
template <class DT, int KB1, int KB2>
__global__ void __launch_bounds__(768) conv1x1_wregd_kernel(const ConvArgs a) {
    constexpr int BM = 64;                     // pixels per step
    constexpr int BNW = 32;                    // channels per consumer wave
    constexpr int BNG = 8 * BNW;               // channels per workgroup
    constexpr int KB = KB1 + KB2;
    constexpr int KS = KB * 4;                 // 16-wide k-slices
    constexpr int XBUF = KB * BM * 128;        // one input tile: KB blocks of [64 px][128 B]
    constexpr int SROW = BNG * 2;              // staging row: 256 packed channels (16-byte chunks XOR-swizzled by the pixel row)
    constexpr int SBUF = BM * SROW;            // one staged output tile (32 KB)
    constexpr int STG_OFF = 2 * XBUF;          // two staging tiles above the two input buffers: 2 x 48 + 2 x 32 KB = all 160 KB
    typedef typename DT::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;

    // work split: workgroup g serves channel slice g % nsl, pixel tiles (g / nsl) + i * (G / nsl); the nsl slices of
    // a pixel tile sit on the same XCD (conv_wreg.hip)
    const int nsl = a.Cout / BNG;
    const int mt = (a.M + BM - 1) / BM;
    const int per = gridDim.x / nsl;
    const int lid = a.no_xcd_map ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int sl = lid % nsl;
    const int tile0 = lid / nsl;
    if (tile0 >= mt) return;
    // The loop is COUNTED in pairs of steps (the two input buffers / register sets alternate at compile time) and has no
    // early exit: a workgroup with an odd number of tiles computes and stores its last tile twice (same values).
    const int n = (mt - tile0 + per - 1) / per;   // pixel tiles of this workgroup (>= 1)
    auto tile_at = [&](int i) { return tile0 + (i < n ? i : n - 1) * per; };
    char* const stg = smem + STG_OFF;

    if (wave >= 8) {
        // ================================ memory waves ================================================================
        const int mtid = tid - 512;                        // 0 .. 255
        const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2_bytes, 0x00020000);
        const uint32_t y_bytes = (uint32_t)((size_t)a.M * a.Cout * 2);
        const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, y_bytes, 0x00020000);
        // input tile image in LDS: block kb (64 channels), pixel row p, 16-byte chunk c at
        //   kb*8192 + p*128 + ((c ^ ((p >> 1) & 7)) << 4)           (conv_igemm's swizzle); staged through registers:
        // a lane carries chunk `sslot` of pixel rows spix and spix + 32, all KB blocks (2 KB registers)
        const int spix = mtid >> 3, sslot = mtid & 7;
        const int sdst = spix * 128 + ((sslot ^ ((spix >> 1) & 7)) << 4);   // (row + 32: same swizzle term, + 4096)
        // (branch-free on purpose: exec-masked regions in the loop cost the compiler its exact vmcnt bookkeeping; the
        // admissibility rule keeps OW > 1, so the divisions need no special case)
        auto load_x = [&](int t, u32x4_t* xr) {
            if (DIR_WREGD_ABL & 8) return;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = t * BM + h * 32 + spix;
                const bool in = m < a.M;
                const uint32_t mm = in ? (uint32_t)m : 0u;
                const uint32_t off1 = (mm * (uint32_t)a.Cin + sslot * 8) * 2;
                const uint32_t base = in ? off1 : kOOBd;
#pragma unroll
                for (int i = 0; i < KB1; ++i) xr[h * KB + i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, base, i * 128, 0);
                // output pixel -> pixel (oh * stride2, ow * stride2) of the second source
                const uint32_t b = __umulhi(mm, a.div_ohw_mul) >> a.div_ohw_shr;
                const uint32_t rem = mm - b * (uint32_t)(a.OH * a.OW);
                const uint32_t oh = __umulhi(rem, a.div_ow_mul) >> a.div_ow_shr;
                const uint32_t ow = rem - oh * (uint32_t)a.OW;
                const uint32_t off2 = (((b * a.H2 + oh * a.stride2) * a.W2 + ow * a.stride2) * a.Cin2 + sslot * 8) * 2;
                const uint32_t base2 = (DIR_WREGD_ABL & 4) ? (in ? (mm * (uint32_t)a.Cin2 + sslot * 8) * 2 : kOOBd) : (in ? off2 : kOOBd);
#pragma unroll
                for (int i = 0; i < KB2; ++i)
                    xr[h * KB + KB1 + i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x2, base2, i * 128, 0);
            }
        };
        auto store_x = [&](const u32x4_t* xr, char* buf) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < KB; ++i) *(u32x4_t*)(buf + i * (BM * 128) + h * (32 * 128) + sdst) = xr[h * KB + i];
        };
        // a tile's stores: 16-byte chunk q = mtid + 256 k -> pixel k * 8 + mtid / 32, chunk mtid % 32
        const int cpix = mtid >> 5, cchunk = mtid & 31;
        const uint32_t ycol = (uint32_t)((sl * BNG) * 2 + cchunk * 16);
        auto store_tile = [&](const char* sb, int m0) {       // rows from m0 on (m0 = M: nothing to store yet)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int pix = k * 8 + cpix;
                const u32x4_t ov = *(const u32x4_t*)(sb + pix * SROW + ((cchunk ^ (pix & 31)) << 4));
                const int m = m0 + pix;
                const uint32_t off = m < a.M ? (uint32_t)m * (uint32_t)(a.Cout * 2) + ycol : kOOBd;
                if (!(DIR_WREGD_ABL & 2)) __builtin_amdgcn_raw_buffer_store_b128(ov, rsrc_y, off, 0, 0);
            }
        };
        auto step = [&](int i, const int cur, u32x4_t* pub, u32x4_t* ld) {
            load_x(tile_at(i + 2), ld);                              // two tiles ahead
            store_tile(stg + (cur ^ 1) * SBUF, i > 0 ? tile_at(i - 1) * BM : a.M);   // the tile staged during the last step
            store_x(pub, smem + (cur ^ 1) * XBUF);                   // requested one step ago; last read one step ago
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (staging reads done, publication in LDS before the hand-off)
            ring_barrier();   // tile i staged, tile i + 1 published
        };
        u32x4_t xa[2 * KB] = {}, xq[2 * KB] = {};
        load_x(tile_at(0), xa);
        store_x(xa, smem);
        load_x(tile_at(1), xa);
        ring_barrier();   // first tile published (and the bias table written)
        int i = 0;
        for (; i < n; i += 2) {
            step(i, 0, xa, xq);
            step(i + 1, 1, xq, xa);
        }
        store_tile(stg + SBUF, tile_at(i - 1) * BM);   // the last staged tile (an odd step's: staging tile 1)
        return;
    }

    // ==================================== consumers ===================================================================
    Ovf<DT> ovf;
    const int n_wave = sl * BNG + wave * BNW;  // first output channel of this wave
    // ---- weights -> registers, once: A-fragment of k-slice ks (rows = this wave's 32 channels) --------------
    frag_t wf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        wf[ks] = *(const DIR_GLOBAL frag_t*)(a.w + (size_t)(n_wave + lrow) * a.Ktot + ks * 16 + 8 * lhi);
    // "already in registers" (conv_wreg.hip: keeps the wait for these loads out of the loop)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wf[ks]));
    // the accumulators of every tile start at the bias of their lane's 16 channels (rows 8 g + 4 lhi + e of the wave's
    // channel tile) - conv_persist.hip's convention, so the sums are the DUAL ring's bit for bit.  The wave's 32 values are
    // wave-uniform: they stay in SCALAR registers (the vector file is full: 96 weight + 32 accumulator registers of 168)
    // and a lane picks its half by lhi.
    float bzs[32];
#pragma unroll
    for (int k = 0; k < 32; ++k)
        bzs[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bias[n_wave + k])));
    const int lswz = (lane >> 1) & 7;
    const int lbase = lrow * 128;
    // this lane's staging writes: pixel j * 32 + lrow, channels wave * 32 + 8 g + 4 lhi .. + 3 (8 bytes) = half of the 16-byte
    // chunk wave * 4 + g, stored at chunk position (wave * 4 + g) ^ lrow of the pixel's row: conflict-free for the 8-byte
    // writes (16 pixel rows per LDS pass) and for the memory waves' 16-byte reads (32 chunks of one row per pass)
    const int swr = lrow * SROW + (((wave * 4) ^ lrow) << 4) + lhi * 8;   // chunk g: swr ^ (g << 4)
    const float floor_v = a.relu ? 0.f : -__builtin_huge_valf();
    auto step = [&](const int cur) {
        const char* xb = smem + cur * XBUF;
        f32x16_t acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float b_lo = bzs[8 * g + e], b_hi = bzs[8 * g + 4 + e];
                    asm volatile("" : "+s"(b_lo), "+s"(b_hi));   // (keeps the select inside the loop: hoisted, it is 16 more live vector registers)
                    acc[j][4 * g + e] = lhi ? b_hi : b_lo;
                }
        // fragment reads run one k-slice ahead of the MFMAs that use them (8 registers), pinned by the scheduling fences
        auto xptr = [&](int ks) { return xb + (ks >> 2) * (BM * 128) + lbase + (((2 * (ks & 3) + lhi) ^ lswz) << 4); };
        frag_t x0n = *(const frag_t*)xptr(0), x1n = *(const frag_t*)(xptr(0) + 32 * 128);
#pragma unroll
        for (int ks = 0; ks < ((DIR_WREGD_ABL & 1) ? 0 : KS); ++ks) {
            const frag_t x0 = x0n, x1 = x1n;
            if (ks + 1 < KS) {
                x0n = *(const frag_t*)xptr(ks + 1);
                x1n = *(const frag_t*)(xptr(ks + 1) + 32 * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = DT::mfma32(wf[ks], x0, acc[0]);
            acc[1] = DT::mfma32(wf[ks], x1, acc[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- ReLU, pack, stage: a lane holds 16 channels of pixel j * 32 + lrow ---------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4] = {acc[j][4 * g + 0], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], floor_v);   // (ReLU as a max with 0 or -inf: no branch per store)
                u32x2_t ov;
                ov[0] = DT::pack(v[0], v[1]);
                ov[1] = DT::pack(v[2], v[3]);
                ovf.see(ov);
                *(u32x2_t*)(stg + cur * SBUF + j * (32 * SROW) + (swr ^ (g << 4))) = ov;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the staged tile has reached LDS before the hand-off)
        ring_barrier();   // tile i is staged, tile i + 1 published
    };
    ring_barrier();   // first tile published (and the bias table written)
    for (int i = 0; i < n; i += 2) {
        step(0);
        step(1);
    }
    ovf.flush(a.ovf);
}

bool conv1x1_wregd_admissible(const ConvArgs& a) {
    return a.x2 && a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.OH && a.W == a.OW && !a.res &&
           a.ksplit <= 1 && a.OW > 1 && a.Cout % 256 == 0 && a.Cin == 128 && a.Cin2 == 256 && a.Ktot == a.Cin + a.Cin2;
}

template <class DT, int KB1, int KB2>
static hipError_t launch_wregd(const ConvArgs& a, hipStream_t stream) {
    constexpr int XBUF = (KB1 + KB2) * 64 * 128;
    constexpr int LDS = 2 * XBUF + 2 * 64 * 256 * 2;
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
    hipLaunchKernelGGL(kern, dim3(per * nsl), dim3(768), LDS, stream, b);
    return hipGetLastError();
}

hipError_t conv1x1_wregd_bf16(const ConvArgs& a, hipStream_t stream) { return launch_wregd<BF16, 2, 4>(a, stream); }
hipError_t conv1x1_wregd_fp16(const ConvArgs& a, hipStream_t stream) { return launch_wregd<FP16, 2, 4>(a, stream); }

}  // namespace dir
