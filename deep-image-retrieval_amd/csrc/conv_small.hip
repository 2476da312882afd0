// conv_small.hip — implicit-GEMM convolution for SMALL maps: 64 x 64 tiles, four consumer waves and four loader waves
// that meet on LDS counters instead of the workgroup barrier (gfx950).
//
// The reference's real regime is batch 1 at native size (dirtorch/test_dir.py:52-55): a layer3 conv of one 1024^2 image is
// 4 096 output pixels.  The tile variants of conv_igemm.hip give such a layer 16-128 workgroups for 256 CUs, and its 64 x 64
// variant - one tile per CU - spends a K-step (four MFMAs per wave) on a workgroup barrier, a counted wait, its share of the
// LDS-DMA issue and a dependent MFMA chain: ~860 cycles per step at any ring depth (profiles/r06_small_map.txt: 16-19 us for
// layer3's 3x3).  This kernel takes those four things apart:
//   * waves 4-7 LOADERS: each issues a quarter of every stage (2 + 2 LDS-DMA instructions of 1 KB per K-step of 64: 64 pixel rows
//     and 64 weight rows of 128 bytes), keeps LAG stages in flight behind a counted vmcnt (nothing but its own LDS-DMA is ever in
//     its queue) and publishes "my part of stage g has landed" in an LDS word; it reuses a ring slot when the four consumers' words
//     say they have left it (NST - LAG - 1 stages of slack between the roles).  Same gather as conv_igemm.hip (any R x S <= 4 x 4,
//     stride, padding; K order = channel slice outermost, taps innermost; chunks of a row XOR-swizzled on the source side).
//   * waves 0-3 CONSUMERS: a 32 x 32 sub-tile each, software-pipelined over the ring: the fragments of stage g + 1 are requested
//     before the MFMAs of stage g are issued, the loaders' words are read only when the cached count does not cover the stage, a
//     slot is given back as soon as its fragment reads are issued; four MFMAs per K-step alternating between TWO accumulators (no
//     dependent chain; summed in the epilogue).  No workgroup barrier anywhere in the loop.
//   * KPS = 1 or 2 K-steps per ring stage (16 / 32 KB: half the hand-offs for the same bytes).
//   * one continuous ring over all the tiles of a persistent workgroup; bias, residual (both requested at the top of a tile)
//     and ReLU in the epilogue, 16-byte stores straight from the accumulators (v_permlane32_swap, conv_ring.hip).
// Measured (profiles/r06_small_map.txt sections 5-6): parity-green, and NOT faster than conv_igemm.hip's 64 x 64 tile with its barrier
// per step - the regime is bounded by L2 bandwidth per FLOP, not by the hand-off.  A tuner candidate (`64x64_small_s8 / _s4 / _s4k2`).
// Sums differ from the other variants in fp32 order (two accumulators, bias last): parity to 16-bit rounding, not bit for bit.
// Every spin is bounded: a wave that waits ~0.5 s raises bit 1 of the overflow word and lets go (results are then garbage,
// the host sees the flag; a lost hand-off must not hang the GPU).
#include "dir_common.h"
#include "conv_igemm.h"

namespace dir {

static constexpr uint32_t kOOBs = 0x80000000u;

__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DIR_LDS void*)lds, 16, voff, soff, 0, 0);
}

// The LOADERS' counter traffic goes through inline assembly: hipcc's waitcnt pass orders every LDS instruction it can see behind
// pending LDS-DMA ("may alias the DMA's destination") with an s_waitcnt vmcnt(0) - one poll or one flag store per stage would drain
// the very queue the loaders exist to keep full.  The counters never alias a ring slot, and the order that matters (stage landed,
// THEN flag) is the counted vmcnt written out below.
__device__ __forceinline__ u32x4_t lds_read4_raw(uint32_t addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write1_raw(uint32_t addr, int v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// KPS = K-steps of 64 per ring stage (one hand-off per stage): 1, or 2 (32 KB stages: half the hand-offs, the same bytes)
template <class DT, int NST, int KPS>
__global__ void __launch_bounds__(512) conv_small_kernel(const ConvArgs a) {
    constexpr int BM = 64, BN = 64;
    constexpr int XS = BM * 128, KSTEP = (BM + BN) * 128, STAGE = KPS * KSTEP;   // per K-step: 8 KB + 8 KB
    // stages a loader keeps in flight behind its counted wait.  It publishes stage c when it issues stage c + LAG, and it may issue
    // that one only when the consumers have left slot (c + LAG) % NST - so NST - LAG - 1 stages of slack decouple the two roles.
    // (LAG = NST - 2, the first form, left ONE: every hand-off then sat on a poll -> issue -> poll round trip, ~0.19 us per stage at
    // any ring depth.)
    constexpr int LAG = NST >= 8 ? 3 : 1;
    constexpr int CNT_OFF = NST * STAGE;                     // [0..3] landed (per loader), [4..7] consumed (per consumer), 16-byte aligned
    constexpr int SPIN_LIMIT = 1 << 23;
    typedef typename DT::frag_t frag_t;
    static_assert(NST >= 3 && 4 * KPS * LAG <= 63, "ring depth");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* const cnt = (int*)(smem + CNT_OFF);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;

    const int ntiles = a.tiles_m * a.tiles_n;
    const int T = a.T / KPS;   // ring stages per tile (the launcher checks T % KPS == 0)
    const int first = xcd_remap(blockIdx.x, gridDim.x);
    if (first >= ntiles) return;
    const int my_tiles = (ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * T;

    if (tid < 8) cnt[tid] = 0;
    __syncthreads();   // the only workgroup barrier of the kernel

    if (wave >= 4) {
        // ================================ loaders ======================================================================
        const int l = wave - 4;
        const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
        // instruction i (0..7) of a panel covers rows 8 i .. 8 i + 7 (8 lanes x 16 B per 128-byte row); this wave issues
        // instructions 2 l and 2 l + 1 of both panels
        int xbase[2];
        uint32_t xmask[2], wvoff[2];
        auto tile_offsets = [&](int tile) __attribute__((always_inline)) {
            const int tile_n = tile % a.tiles_n, tile_m = tile / a.tiles_n;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (2 * l + i) * 8 + (lane >> 3);
                const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                const int m = tile_m * BM + row;
                const bool mvalid = m < a.M;
                const uint32_t mm = mvalid ? (uint32_t)m : 0u;
                const uint32_t b = a.div_ohw_mul ? (__umulhi(mm, a.div_ohw_mul) >> a.div_ohw_shr) : mm;
                const uint32_t rem = mm - b * (uint32_t)(a.OH * a.OW);
                const uint32_t oh = a.div_ow_mul ? (__umulhi(rem, a.div_ow_mul) >> a.div_ow_shr) : rem;
                const uint32_t ow = rem - oh * (uint32_t)a.OW;
                const int ih0 = (int)oh * a.stride - a.pad, iw0 = (int)ow * a.stride - a.pad;
                xbase[i] = (((int)b * a.H + ih0) * a.W + iw0) * a.Cin * 2 + chunk * 16;
                // validity bit per tap (conv_igemm.hip): rows r in [max(0, -ih0), min(R, H - ih0)), columns likewise
                auto range_bits = [](int lo, int hi) -> uint32_t { return hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u; };
                const uint32_t rbits = range_bits(max(0, -ih0), min(a.R, a.H - ih0));
                const uint32_t cbits = range_bits(max(0, -iw0), min(a.S, a.W - iw0));
                uint32_t mask = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) mask |= ((rbits >> r) & 1u) ? (cbits << (r * a.S)) : 0u;
                xmask[i] = mvalid ? mask : 0u;
                wvoff[i] = (uint32_t)(((tile_n * BN + row) * a.Ktot + chunk * 8) * 2);
            }
        };
        int tile = first, t = 0, tap = 0, cc = 0, r = 0, s = 0, slot = 0;
        tile_offsets(first);
        const uint32_t cnt_lds = (uint32_t)(uintptr_t)(DIR_LDS int*)cnt;
        for (int g = 0; g < total; ++g) {
            if (g >= NST) {   // the consumers have left the slot this stage goes to (bounded spin)
                for (int spins = 0;; ++spins) {
                    const u32x4_t c4 = lds_read4_raw(cnt_lds + 16);
                    if ((int)min(min(c4[0], c4[1]), min(c4[2], c4[3])) >= g + 1 - NST) break;
                    if (spins > SPIN_LIMIT) {
                        if (a.ovf && lane == 0) atomicOr(a.ovf, 2);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
#pragma unroll
            for (int kk = 0; kk < KPS; ++kk) {
                char* dst = smem + slot * STAGE + kk * KSTEP + (2 * l) * 1024;
                const int koff = ((r * a.W + s) * a.Cin + cc * 64) * 2;
                const int wstep = tap * (a.Cin / 64) + cc;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint32_t v = ((xmask[i] >> tap) & 1u) ? (uint32_t)(xbase[i] + koff) : kOOBs;
                    dma16s(rsrc_x, dst + i * 1024, v, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) dma16s(rsrc_w, dst + XS + i * 1024, wvoff[i], wstep * 128);
                // K order: channel slice outermost, taps innermost (conv_igemm.hip)
                ++tap;
                if (++s == a.S) {
                    s = 0;
                    if (++r == a.R) {
                        r = 0;
                        tap = 0;
                        ++cc;
                    }
                }
            }
            slot = slot + 1 == NST ? 0 : slot + 1;
            if (++t == T) {
                t = 0, tap = 0, cc = 0, r = 0, s = 0;
                tile += (int)gridDim.x;
                if (tile < ntiles) tile_offsets(tile);
            }
            if (g >= LAG) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(4 * KPS * LAG) : "memory");   // my part of stage g - LAG has landed
                if (lane == 0) lds_write1_raw(cnt_lds + 4 * l, g - LAG + 1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) lds_write1_raw(cnt_lds + 4 * l, total);
        return;
    }

    // ==================================== consumers ===================================================================
    const int cn = wave & 1, cm = wave >> 1;
    const int lswz = (lane >> 1) & 7;
    int loff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) loff[ks] = lrow * 128 + (((2 * ks + lhi) ^ lswz) << 4);
    const int xfrag = (cm * 32) * 128;
    const int wfrag = XS + (cn * 32) * 128;
    Ovf<DT> ovf;
    // The consumer's stream is software-pipelined over the ring (profiles/r06_small_map.txt: with poll -> fragment reads -> MFMAs ->
    // flag strictly in sequence a hand-off cost ~0.19 us, more than the stage's bytes): the fragments of stage g + 1 are requested
    // before the MFMAs of stage g are issued (two register sets), the loaders' words are read only when the cached count does not
    // cover the stage, and a slot is given back as soon as its fragment reads have been ISSUED - LDS executes a wave's
    // instructions in order, so the flag write cannot pass them, and the loader's LDS-DMA into the slot is issued only after it
    // has seen the flag.
    int landed = 0;   // stages every loader is known to have landed
    auto ensure = [&](int need) {
        if (landed >= need) return;
        for (int spins = 0;; ++spins) {
            const int v0 = __hip_atomic_load(cnt + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int v1 = __hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int v2 = __hip_atomic_load(cnt + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int v3 = __hip_atomic_load(cnt + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            landed = min(min(v0, v1), min(v2, v3));
            if (landed >= need) break;
            if (spins > SPIN_LIMIT) {
                if (a.ovf && lane == 0) atomicOr(a.ovf, 2);
                landed = need;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    frag_t wfA[4 * KPS], xfA[4 * KPS], wfB[4 * KPS], xfB[4 * KPS];
    auto read_frags = [&](int slot_, frag_t* wf, frag_t* xf) {
#pragma unroll
        for (int kk = 0; kk < KPS; ++kk) {
            const char* stage = smem + slot_ * STAGE + kk * KSTEP;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                wf[kk * 4 + ks] = *(const frag_t*)(stage + wfrag + loff[ks]);
                xf[kk * 4 + ks] = *(const frag_t*)(stage + xfrag + loff[ks]);
            }
        }
    };
    int tile = first, t = 0;
    f32x16_t acc0, acc1;
    f32x4_t b4[4];
    u32x4_t rres[2] = {};
    int n_wave = 0, m = 0;
    bool mok = false;
    auto tile_begin = [&]() {
        const int tile_n = tile % a.tiles_n, tile_m = tile / a.tiles_n;
        n_wave = tile_n * BN + cn * 32;
        m = tile_m * BM + cm * 32 + lrow;
        mok = m < a.M;
        // requested now, used in the epilogue: the bias of this lane's 16 channels and its two residual pieces
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) b4[gg] = *(const DIR_GLOBAL f32x4_t*)(a.bias + n_wave + 8 * gg + 4 * lhi);
        if (a.res && mok) {
#pragma unroll
            for (int h = 0; h < 2; ++h) rres[h] = gload16(a.res + ((size_t)m * a.Cout + n_wave + h * 16 + lhi * 8));
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc0[e] = 0.f, acc1[e] = 0.f;
    };
    auto tile_end = [&]() {
        // ---- epilogue: the two partial sums + bias (+ residual), ReLU, pack; v_permlane32_swap pairs the half-waves' 8-byte
        //      pieces into 16-byte stores.  acc[4 gg + e] = channel 8 gg + 4 lhi + e of pixel lrow.
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // Residual pieces are laid out like the stores (8 consecutive channels per lane), the accumulators like the MFMA
            // (4 + 4 channels split over the half-waves): bring the accumulators to the store layout in fp32 first.
            float lo4[4], hi4[4];   // this lane's values of group 2h (q = 0) and 2h + 1 (q = 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo4[e] = (acc0[4 * (2 * h) + e] + acc1[4 * (2 * h) + e]) + b4[2 * h][e];
                hi4[e] = (acc0[4 * (2 * h + 1) + e] + acc1[4 * (2 * h + 1) + e]) + b4[2 * h + 1][e];
            }
            float v8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // after the swap lanes 0-31 hold (q = 0 of lhi 0, q = 0 of lhi 1) = channels 8 (2h) .. + 7, lanes 32-63 the q = 1 pair
                const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, lo4[e]), __builtin_bit_cast(uint32_t, hi4[e]), false, false);
                v8[e] = __builtin_bit_cast(float, (uint32_t)r[0]);
                v8[4 + e] = __builtin_bit_cast(float, (uint32_t)r[1]);
            }
            if (a.res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float lo, hi;
                    DT::unpack(rres[h][e], lo, hi);
                    v8[2 * e] += lo;
                    v8[2 * e + 1] += hi;
                }
            }
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = fmaxf(v8[e], 0.f);
            }
            u32x4_t ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = DT::pack(v8[2 * e], v8[2 * e + 1]);
            if (mok) {
                gstore16(a.y + ((size_t)m * a.Cout + n_wave + h * 16 + lhi * 8), ov);
                ovf.see(ov);
            }
        }
    };
    // one step = stage g from register set `cur`, stage g + 1 requested into `nxt`
    auto step = [&](int g, frag_t* wfc, frag_t* xfc, frag_t* wfn, frag_t* xfn) {
        // stage g's fragment reads were issued one step ago: its slot goes back now (the empty asm keeps the compiler from sinking
        // those reads below the flag; the hardware executes this wave's LDS instructions in order)
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_store(cnt + 4 + wave, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        if (g + 1 < total) {
            ensure(g + 2);
            read_frags((g + 1) % NST, wfn, xfn);
        }
        if (t == 0) tile_begin();
#pragma unroll
        for (int kk = 0; kk < KPS; ++kk) {
            acc0 = DT::mfma32(wfc[kk * 4 + 0], xfc[kk * 4 + 0], acc0);
            acc1 = DT::mfma32(wfc[kk * 4 + 1], xfc[kk * 4 + 1], acc1);
            acc0 = DT::mfma32(wfc[kk * 4 + 2], xfc[kk * 4 + 2], acc0);
            acc1 = DT::mfma32(wfc[kk * 4 + 3], xfc[kk * 4 + 3], acc1);
        }
        if (++t == T) {
            tile_end();
            t = 0;
            tile += (int)gridDim.x;
        }
    };
    ensure(1);
    read_frags(0, wfA, xfA);
    for (int g = 0; g < total; g += 2) {
        step(g, wfA, xfA, wfB, xfB);
        if (g + 1 < total) step(g + 1, wfB, xfB, wfA, xfA);
    }
    ovf.flush(a.ovf);
}

bool conv_small_admissible(const ConvArgs& a) {
    return a.x2 == nullptr && a.Cin % 64 == 0 && a.Cout % 64 == 0 && a.R >= 1 && a.R <= 4 && a.S >= 1 && a.S <= 4 && a.ksplit <= 1 &&
           (long)a.B * a.H * a.W * a.Cin < (1L << 30) && (long)a.M * a.Cout < (1L << 30);
}

template <class DT, int NST, int KPS = 1>
static hipError_t launch_small(const ConvArgs& a, hipStream_t stream) {
    constexpr int LDS = NST * KPS * 128 * 128 + 64;
    static_assert(LDS <= 160 * 1024, "LDS map");
    if ((a.Ktot / 64) % KPS != 0) return hipErrorInvalidValue;
    auto kern = conv_small_kernel<DT, NST, KPS>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_done); e != hipSuccess) return e;
    ConvArgs b = a;
    b.T = a.Ktot / 64;
    b.tiles_m = ceil_div(a.M, 64);
    b.tiles_n = a.Cout / 64;
    b.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 2);
    b.w_bytes = (uint32_t)((size_t)a.Cout * a.Ktot * 2);
    auto fd = [](uint32_t d, uint32_t& mul, uint32_t& shr) {   // exact n / d for n < 2^31 (conv_igemm.hip's constants; d <= 1: mul = 0)
        if (d <= 1) { mul = 0; shr = 0; return; }
        uint32_t l = 0;
        while ((1ull << l) < d) ++l;
        mul = (uint32_t)(((1ull << (31 + l)) + d - 1) / d);
        shr = l - 1;
    };
    fd((uint32_t)(a.OH * a.OW), b.div_ohw_mul, b.div_ohw_shr);
    fd((uint32_t)a.OW, b.div_ow_mul, b.div_ow_shr);
    const int ntiles = b.tiles_m * b.tiles_n;
    const int slots = cu_count() * (LDS <= 80 * 1024 ? 2 : 1);
    hipLaunchKernelGGL(kern, dim3(ntiles < slots ? ntiles : slots), dim3(512), LDS, stream, b);
    return hipGetLastError();
}

hipError_t conv_small_launch(const ConvArgs& a, int dtype, int nst, hipStream_t stream) {
    if (nst == 4) return dtype == DIR_BF16 ? launch_small<BF16, 4>(a, stream) : launch_small<FP16, 4>(a, stream);
    if (nst == 5) return dtype == DIR_BF16 ? launch_small<BF16, 4, 2>(a, stream) : launch_small<FP16, 4, 2>(a, stream);   // (table: stages 5 = 4 slots x 2 K-steps)
    return dtype == DIR_BF16 ? launch_small<BF16, 8>(a, stream) : launch_small<FP16, 8>(a, stream);
}

}  // namespace dir
