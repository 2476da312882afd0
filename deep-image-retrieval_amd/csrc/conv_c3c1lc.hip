// conv_c3c1lc.hip — the DS seam of layer1's first block (conv_c3c1.hip's DS form) with the work split by wave ROLE (gfx950).
//
//     out  = relu([W3 | Wds] . [t2 ; x] + b3 + bds)       [M, 256]   conv3 + bn3 + the downsample branch + add + ReLU
//     t1'  = relu(W1' . out + b1')                          [M, 64]    the next block's conv1
// (dirtorch/nets/backbones/resnet.py:78-85 with :134-141, then :70-72).  Same arithmetic, same MFMA order per accumulator,
// same 16-bit values handed from the first GEMM to the second as conv_c3c1_kernel<.., DS = true, ..>: the two agree bit for
// bit (tests/test_ops_gpu.py).  What differs is who does what.  In the one-role kernel every wave loads, multiplies and
// stores, and its 56 MFMAs per 64-pixel tile (paired weights: two per product term) sit between the barriers of a loop that
// is otherwise memory-bound: 0.52-0.55 ms at batch 32 where the bytes alone would take 0.42 (4.0 TB/s against the 5.2 of the
// plain seams; round-5 review, item 2).  Here
//   waves 0-7  CONSUMERS  hold both weight matrices in registers, multiply, and hand BOTH outputs to LDS as packed 16-bit
//              tiles (the 256-wide tile in the second GEMM's B-operand layout, where they read it back);
//              no global memory instruction inside the loop;
//   waves 8-11 MEMORY waves request the input tiles two steps ahead (registers), publish them to LDS, and store the two
//              output tiles of the PREVIOUS step from LDS (512-byte and 128-byte pixel rows) while the consumers multiply
//              the next one.  They never multiply.
// ONE barrier per tile joins the roles (both outputs of tile i staged, input tile i + 1 published).  The consumers' own
// rendezvous - the 256-wide tile is complete, the K halves of the second GEMM are visible - run on LDS counters the memory
// waves never wait on (a first form with three workgroup barriers per tile, which parks the memory waves through the second
// GEMM, measured the same: both take 10-11 % off the one-role kernel inside the network, 0.553 -> 0.493 and 0.519 -> 0.468 ms on
// two boxes, gpurun_out/r6c3c1lc*); the 256-wide output tile is double-buffered so that its stores overlap the next tile's first
// GEMM, the 64-wide one is guarded by a "consumed" counter.
// The structure conv_wregd.hip arrived at, with its measurements: profiles/r06_wregd.txt.
#include "dir_common.h"
#include "conv_igemm.h"

namespace dir {

static constexpr uint32_t kOOBl = 0x80000000u;

// WP: DIR_FP16P - the weights are fp16 PAIRS (a.w_lo, a.w2_lo) and the block input comes with its lo plane (a.x2_lo), staged as
// a third 64-channel K block against the downsample's hi weights (conv_c3c1.hip WP3 / WP1)
template <class DT, bool WP>
__global__ void __launch_bounds__(768) conv_c3c1ds_lc_kernel(const ConvArgs a) {
    constexpr int KW = 128;                      // contraction length of [W3 | Wds]
    constexpr int KA = KW + (WP ? 64 : 0);       // ... of the staged tile (+ the block input's lo plane)
    constexpr int C4 = 256, P2 = 64, BM = 64;
    constexpr int KSA = KA / 16, KSW = KW / 16;  // first GEMM: k-slices of the tile / with weights of their own
    constexpr int NXB = KA / 64;                 // 64-channel blocks of the staged tile
    constexpr int KSB = (C4 / 2) / 16;           // second GEMM: k-slices per K half
    constexpr int XBUF = BM * KA * 2;            // one input tile: NXB blocks of [64 px][128 B]
    constexpr int OUTB = BM * C4 * 2;            // the 256-wide output tile: 4 blocks of [64 px][128 B]
    constexpr int T1B = BM * P2 * 2;             // the 64-wide output tile: [64 px][128 B]
    constexpr int EROW = 32 * 4 + 16;            // K-half exchange row: 32 fp32 + pad
    // paired form: the lo plane of W1' lives in LDS as ready-made MFMA fragments (32 KB; with it in registers the consumers spill),
    // and the K-half exchange rows alias the input buffer the first GEMM has just finished with (free until the step's last barrier)
    constexpr int W1L_BYTES = WP ? 2 * 2 * KSB * 64 * 16 : 0;
    constexpr int OUT_OFF = 2 * XBUF, T1_OFF = OUT_OFF + 2 * OUTB, W1L_OFF = T1_OFF + T1B, EX_OFF = W1L_OFF + W1L_BYTES,
                  BIAS_OFF = EX_OFF + (WP ? 0 : 4 * 32 * EROW), CNT_OFF = BIAS_OFF + (C4 + P2) * 4;
    static_assert(!WP || 4 * 32 * EROW <= XBUF, "exchange rows fit an input buffer");
    typedef typename DT::frag_t frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;

    const int mt = (a.M + BM - 1) / BM;
    const int per = gridDim.x;
    const int tile0 = blockIdx.x;
    if (tile0 >= mt) return;
    // counted in PAIRS of steps, no early exit (conv_wregd.hip): an odd tile count computes and stores the last tile twice
    const int n = (mt - tile0 + per - 1) / per;
    auto tile_at = [&](int i) { return tile0 + (i < n ? i : n - 1) * per; };
    float* const sbias3 = (float*)(smem + BIAS_OFF);
    float* const sbias1 = sbias3 + C4;
    // LDS counters (monotonic, one arrival per wave and step): [0] consumers past their first GEMM's stores, [1..4] the K-half
    // giver of role pair (wave & 3), [5] memory waves done reading the staged 64-wide tile
    int* const cnt = (int*)(smem + CNT_OFF);
    auto arrive = [&](int* c) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // (every spin is bounded: a wave that waits ~0.3 s raises bit 1 of the overflow word and lets go - the results are then garbage
    // and the host sees the flag; a lost hand-off must not hang the GPU)
    auto await = [&](int* c, int target) {
        for (int spins = 0; __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target; ++spins) {
            if (spins > (1 << 22)) {
                if (a.ovf && lane == 0) atomicOr(a.ovf, 2);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };

    if (wave >= 8) {
        // ================================ memory waves ================================================================
        const int mtid = tid - 512;                        // 0 .. 255
        const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (uint32_t)((size_t)a.M * 128), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, (uint32_t)((size_t)a.M * 128), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_x2l =
            __builtin_amdgcn_make_buffer_rsrc((void*)(WP ? a.x2_lo : a.x2), 0, (uint32_t)((size_t)a.M * 128), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (uint32_t)((size_t)a.M * C4 * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_y2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.y2, 0, (uint32_t)((size_t)a.M * P2 * 2), 0x00020000);
        sbias3[mtid] = a.bias[mtid];
        if (mtid < P2) sbias1[mtid] = a.bias2[mtid];
        if (mtid < 8) cnt[mtid] = 0;
        // input tile image in LDS: block kb (64 channels), pixel row p, 16-byte chunk c at kb*8192 + p*128 + ((c ^ ((p >> 1) & 7)) << 4);
        // a lane carries chunk `sslot` of pixel rows spix and spix + 32 of every block
        const int spix = mtid >> 3, sslot = mtid & 7;
        const int sdst = spix * 128 + ((sslot ^ ((spix >> 1) & 7)) << 4);   // (row + 32: same swizzle term, + 4096)
        auto load_x = [&](int t, u32x4_t* xr) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = t * BM + h * 32 + spix;
                const uint32_t base = m < a.M ? (uint32_t)((m * 64 + sslot * 8) * 2) : kOOBl;
                xr[h * NXB + 0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, base, 0, 0);
                xr[h * NXB + 1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x2, base, 0, 0);
                if (WP) xr[h * NXB + NXB - 1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x2l, base, 0, 0);
            }
        };
        auto store_x = [&](const u32x4_t* xr, char* buf) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < NXB; ++i) *(u32x4_t*)(buf + i * (BM * 128) + h * (32 * 128) + sdst) = xr[h * NXB + i];
        };
        // the 256-wide tile: per pass a wave stores two pixel rows (4 blocks x 128 B each); 16 consecutive lanes read the 8
        // chunks of one block for two neighbouring pixels - conflict-free in LDS, 128-byte runs in memory
        const int oc8 = mtid & 7, opar = (mtid >> 3) & 1, oblk = (mtid >> 4) & 3, owv = mtid >> 6;
        auto store_out = [&](const char* ob, int m0) {     // rows from m0 on (m0 = M: nothing to store yet)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int pix = 2 * (k * 4 + owv) + opar;
                const u32x4_t ov = *(const u32x4_t*)(ob + oblk * (BM * 128) + pix * 128 + ((oc8 ^ ((pix >> 1) & 7)) << 4));
                const int m = m0 + pix;
                const uint32_t off = m < a.M ? (uint32_t)m * (uint32_t)(C4 * 2) + (uint32_t)(oblk * 128 + oc8 * 16) : kOOBl;
                __builtin_amdgcn_raw_buffer_store_b128(ov, rsrc_y, off, 0, 0);
            }
        };
        auto store_t1 = [&](int m0) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int pix = k * 32 + (mtid >> 3);
                const u32x4_t ov = *(const u32x4_t*)(smem + T1_OFF + pix * 128 + ((oc8 ^ ((pix >> 1) & 7)) << 4));
                const int m = m0 + pix;
                const uint32_t off = m < a.M ? (uint32_t)m * (uint32_t)(P2 * 2) + (uint32_t)(oc8 * 16) : kOOBl;
                __builtin_amdgcn_raw_buffer_store_b128(ov, rsrc_y2, off, 0, 0);
            }
        };
        auto step = [&](int i, const int cur, u32x4_t* pub, u32x4_t* ld) {
            load_x(tile_at(i + 2), ld);                              // two tiles ahead
            const int m_prev = i > 0 ? tile_at(i - 1) * BM : a.M;
            store_t1(m_prev);                                        // the tiles staged during the last step: the single-buffered one
            arrive(cnt + 5);                                         // first (its reads are done: the consumers may stage the next)
            store_out(smem + OUT_OFF + (cur ^ 1) * OUTB, m_prev);
            store_x(pub, smem + (cur ^ 1) * XBUF);                   // requested one step ago; last read one step ago
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (staging reads done, publication in LDS before the hand-off)
            ring_barrier();   // both outputs of step i staged, tile i + 1 published
        };
        u32x4_t xa[2 * NXB] = {}, xq[2 * NXB] = {};
        load_x(tile_at(0), xa);
        store_x(xa, smem);
        load_x(tile_at(1), xa);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ring_barrier();   // first tile published, bias tables written
        int i = 0;
        for (; i < n; i += 2) {
            step(i, 0, xa, xq);
            step(i + 1, 1, xq, xa);
        }
        store_out(smem + OUT_OFF + OUTB, tile_at(i - 1) * BM);   // the last step's tiles (an odd step: buffer 1)
        store_t1(tile_at(i - 1) * BM);
        return;
    }

    // ==================================== consumers ===================================================================
    Ovf<DT> ovf;
    const int n_wave = wave * 32;               // first GEMM: this wave's 32 output channels
    // second GEMM roles (conv_c3c1.hip, P2 = 64): n-tile = w & 1, pixel strip = (w >> 1) & 1, K half = w >> 2
    const int nt = wave & 1, jb = (wave >> 1) & 1, kh = wave >> 2;
    frag_t w3[KSW], w3l[WP ? KSW : 1], w1[KSB];
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
        w3[ks] = *(const DIR_GLOBAL frag_t*)(a.w + (size_t)(n_wave + lrow) * KW + ks * 16 + 8 * lhi);
        if (WP) w3l[ks] = *(const DIR_GLOBAL frag_t*)(a.w_lo + (size_t)(n_wave + lrow) * KW + ks * 16 + 8 * lhi);
    }
#pragma unroll
    for (int ks = 0; ks < KSB; ++ks) {
        w1[ks] = *(const DIR_GLOBAL frag_t*)(a.w2 + (size_t)(nt * 32 + lrow) * C4 + kh * (C4 / 2) + ks * 16 + 8 * lhi);
        // lo plane -> LDS, fragment image [n-tile][K half][k-slice][lane] (written by the jb = 0 wave of each role pair)
        if (WP && jb == 0)
            *(frag_t*)(smem + W1L_OFF + (((nt * 2 + kh) * KSB + ks) * 64 + lane) * 16) =
                *(const DIR_GLOBAL frag_t*)(a.w2_lo + (size_t)(nt * 32 + lrow) * C4 + kh * (C4 / 2) + ks * 16 + 8 * lhi);
    }
    const char* const w1lf = smem + W1L_OFF + ((nt * 2 + kh) * KSB * 64 + lane) * 16;
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
        asm volatile("" : "+v"(w3[ks]));
        if (WP) asm volatile("" : "+v"(w3l[ks]));
    }
#pragma unroll
    for (int ks = 0; ks < KSB; ++ks) asm volatile("" : "+v"(w1[ks]));
    const int lswz = (lane >> 1) & 7;
    const int lbase = lrow * 128;
    // a lane's 16 accumulator values are channels 8 g + 4 lhi + e (g, e = 0..3) of pixel lrow: four 8-byte pieces per
    // strip, written at the B-operand position of (channel chunk, pixel) - chunk ^ ((pixel >> 1) & 7)
    const int exo = (wave & 3) * (32 * EROW) + lrow * EROW + (4 * lhi) * 4;
    int it = 0;   // steps done
    auto step = [&](const int cur) {
        const char* xb = smem + cur * XBUF;
        char* const exw = smem + (WP ? cur * XBUF : EX_OFF) + exo;
        char* const ob = smem + OUT_OFF + cur * OUTB;
        // ================= first GEMM: out = relu([t2 ; x] . [W3 | Wds]^T + b3 + bds) ==================================
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x16_t acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSA; ++ks) {
                const frag_t xf = *(const frag_t*)(xb + (ks >> 2) * (BM * 128) + j * (32 * 128) + lbase +
                                                   (((2 * (ks & 3) + lhi) ^ lswz) << 4));
                // (K block 2 of the paired form = the block input's lo plane: the downsample's hi weights once more)
                const int kw = ks < KSW ? ks : ks - 4;
                acc = DT::mfma32(w3[kw], xf, acc);
                if (WP && ks < KSW) acc = DT::mfma32(w3l[WP ? ks : 0], xf, acc);
                if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            const int p = j * 32 + lrow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nch = n_wave + 8 * g + 4 * lhi;
                const f32x4_t b4 = *(const f32x4_t*)(sbias3 + nch);
                float v[4] = {acc[4 * g + 0] + b4[0], acc[4 * g + 1] + b4[1], acc[4 * g + 2] + b4[2], acc[4 * g + 3] + b4[3]};
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                u32x2_t ov;
                ov[0] = DT::pack(v[0], v[1]);
                ov[1] = DT::pack(v[2], v[3]);
                ovf.see(ov);
                *(u32x2_t*)(ob + (nch >> 6) * (BM * 128) + p * 128 + ((((nch & 63) >> 3) ^ ((p >> 1) & 7)) << 4) + (nch & 4) * 2) = ov;
            }
        }
        arrive(cnt);
        await(cnt, 8 * (it + 1));   // the 256-wide tile is complete (all eight consumers have stored their channels)

        // ================= second GEMM: t1' = relu(out . W1'^T + b1') ====================================================
        f32x16_t acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
            const int kb = kh * (C4 / 128) + (ks >> 2);
            const frag_t of = *(const frag_t*)(ob + kb * (BM * 128) + jb * (32 * 128) + lbase + (((2 * (ks & 3) + lhi) ^ lswz) << 4));
            acc1 = DT::mfma32(w1[ks], of, acc1);
            if (WP) acc1 = DT::mfma32(*(const frag_t*)(w1lf + ks * (64 * 16)), of, acc1);
            if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        // the K halves meet in fp32 (fixed order: half 0 + half 1): the kh = 1 wave hands its partial to wave w ^ 4, whose
        // lanes hold the same (channel, pixel) positions
        if (kh == 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t v = {acc1[4 * g + 0], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
                *(f32x4_t*)(exw + g * 32) = v;
            }
        }
        if (kh == 1) arrive(cnt + 1 + (wave & 3));
        if (kh == 0) {
            await(cnt + 1 + (wave & 3), it + 1);   // the partner's partial is visible
            await(cnt + 5, 4 * (it + 1));          // ... and the memory waves have read the previous 64-wide tile
            const int p = jb * 32 + lrow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t h1 = *(const f32x4_t*)(exw + g * 32);
                const int nch = nt * 32 + 8 * g + 4 * lhi;
                const f32x4_t b4 = *(const f32x4_t*)(sbias1 + nch);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (acc1[4 * g + e] + h1[e]) + b4[e];
                if (a.relu2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                u32x2_t ov;
                ov[0] = DT::pack(v[0], v[1]);
                ov[1] = DT::pack(v[2], v[3]);
                ovf.see(ov);
                *(u32x2_t*)(smem + T1_OFF + p * 128 + (((nch >> 3) ^ ((p >> 1) & 7)) << 4) + (nch & 4) * 2) = ov;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ++it;
        ring_barrier();   // both outputs staged, next input tile published
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the lo-plane fragments have reached LDS)
    ring_barrier();   // first tile published, bias tables written
    for (int i = 0; i < n; i += 2) {
        step(0);
        step(1);
    }
    ovf.flush(a.ovf);
}

bool conv_c3c1ds_lc_admissible(const ConvArgs& a) {
    // the DS form of conv_c3c1_admissible (layer1's first block), paired weights or not
    return a.x2 != nullptr && a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.OH && a.W == a.OW && a.Cin == 64 &&
           a.Cin2 == 64 && a.Cout == 256 && a.res == nullptr && a.w2 != nullptr && a.bias2 != nullptr && a.y2 != nullptr &&
           a.Cout2 == 64 && (long)a.M * a.Cout < (1L << 30) &&
           ((!a.w_lo && !a.w2_lo && !a.x2_lo) || (a.w_lo && a.w2_lo && a.x2_lo));
}

template <class DT, bool WP>
static hipError_t launch_c3c1ds_lc(const ConvArgs& a, hipStream_t stream) {
    constexpr int KA = 128 + (WP ? 64 : 0);
    constexpr int LDS = 2 * 64 * KA * 2 + 2 * 64 * 256 * 2 + 64 * 64 * 2 + (WP ? 2 * 2 * 8 * 64 * 16 : 4 * 32 * (32 * 4 + 16)) + (256 + 64) * 4 + 32;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv_c3c1ds_lc_kernel<DT, WP>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    const int mt = (a.M + 63) / 64;
    const int ncu = cu_count();
    hipLaunchKernelGGL(kern, dim3(mt < ncu ? mt : ncu), dim3(768), LDS, stream, a);
    return hipGetLastError();
}

hipError_t conv_c3c1ds_lc_launch(const ConvArgs& a, int dtype, hipStream_t stream) {
    if (a.w_lo) return dtype == DIR_FP16 ? launch_c3c1ds_lc<FP16, true>(a, stream) : hipErrorInvalidValue;
    return dtype == DIR_BF16 ? launch_c3c1ds_lc<BF16, false>(a, stream) : launch_c3c1ds_lc<FP16, false>(a, stream);
}

}  // namespace dir
