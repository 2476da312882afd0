// dir_common.h — shared device/host helpers for the gfx950 descriptor engine.
// Written for MI355X (CDNA4) only: 64-lane wavefronts, MFMA, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <string>

#include "../../include/dir_engine.h"

namespace dir {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

#define DIR_GLOBAL __attribute__((address_space(1)))
#define DIR_LDS __attribute__((address_space(3)))

// ---- thread-local error string ----------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define DIR_HIP_CHECK(expr)                                                                \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return ::dir::fail(DIR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// ---- A/B switches --------------------------------------------------------------------------------------
// Every DIRTORCH_AMD_* switch the library honours.  The environment is read ONCE (first use) and again only when the
// host calls dir_reload_env() - what a test or an A/B script does after flipping a variable inside one process; an
// engine copies the switches that steer its forward at dir_engine_create (`dir_engine::sw`), the kernel pickers and the
// per-op entry points read the current process-wide snapshot (include/dir_engine.h lists which is which).  No launch
// path calls getenv.
struct Env {
    bool c3c1_off = false, c3c1_force = false;   // DIRTORCH_AMD_C3C1 = 0 | force: the fused conv3 -> conv1 seam kernels
    bool no_ds_seam = false, no_dual = false;    // ..._NO_DS_SEAM, ..._NO_DUAL: the downsample as its own launch again
    bool rev_conv1 = false, rev_conv3 = false;   // ..._REV_CONV1 / _REV_CONV3: reversed tile order (measured, no gain)
    bool unfused_stem = false, stem_v1 = false;  // ..._UNFUSED_STEM, ..._STEM_V1: conv + maxpool apart / one tile per workgroup
    bool no_patchlc = false, no_wreg = false, no_patchw = false, no_x3 = false, no_patchs = false;   // heuristic: skip a kernel
    bool patchw_pack = false, no_patchw_pack = false;   // ..._PATCHW_PACK: per-op conv_patchw.hip launches pack the filter into stage images (stream-ordered scratch); ..._NO_PATCHW_PACK: the engine's launches gather from the [Cout][3][3][Cin] layout again
    bool small_k2 = false;                       // ..._SMALL_K2: small maps with >= 192 tiles of 64 x 64 on conv_small.hip's 128 KB two-K-steps-per-stage tile too (one-stream callers: +8 %; two to four streams: -3 %)
    bool no_patchs2 = false;                     // ..._NO_PATCHS2: the strided 3x3 convs on conv_igemm.hip's generic tiles instead of conv_patchs2.hip
    bool x3_k2048 = false;                       // ..._X3_K2048: the deep-X 1x1 ring only from K = 2048 (the picker's rule before round 6) instead of from K = 1024 for 256-channel outputs
    bool lc1x1 = false;                          // ..._LC1X1 (opt-in, A/B): conv_persistlc.hip (loader / consumer 256 x 256 ring) wherever conv_persist.hip's residual-free and two-source forms run
    bool no_c3c1lc = false;                      // ..._NO_C3C1LC: layer1's DS seam on conv_c3c1.hip's one-role kernel instead of conv_c3c1lc.hip
    bool no_smallmap = false;                    // ..._NO_SMALLMAP: the kernel picker without round 6's small-map rules (64x128_w2x2 / 64x64_w2x2_s8)
    bool no_wregd = false;                       // ..._NO_WREGD: layer2's two-source GEMM on conv_persist.hip's DUAL ring again instead of conv_wregd.hip
    bool no_patchw_lc = false;                   // ..._NO_PATCHW_LC: conv_patchw.hip's one-role kernel instead of its loader / consumer form
    bool no_xcdmap = false;                      // ..._NO_XCDMAP: plain tile order in the persistent 1x1 kernels
    bool no_pair_patch = false, pair_acts = false;   // DIR_FP16P: ..._NO_PAIR_PATCH, ..._PAIR_ACTS
    int pair_stages = 1;                         // ..._PAIR_STAGES = 1..4 (validated at finalize)
    bool sim_v1 = false, sim_exact = false;      // ..._SIM_V1 (one-role kernel), ..._SIM_EXACT (fp32 MFMA chain at any size)
    bool experiments = false;                    // ..._EXPERIMENTS: opt-in kernels of an experiments build (conv_ring / conv_seam3)
    bool no_inplace = false;                     // ..._NO_INPLACE: layers 3-4's identity blocks ping-pong again instead of writing their output in place
    bool no_stem_u8 = false;                     // ..._NO_STEM_U8: DIR_FP16P on the uint8 feed takes the generic paired stem (image pair, three MFMAs per term) again
    bool stem_pair_old = false;                  // ..._STEM_PAIR_OLD: the generic paired stem as conv_pair.hip's stem_pool_pair_persist_kernel (3 x 15 pooled tiles through a 64 KB fp32 conv tile)
    bool stem_u8_prep = false;                   // ..._STEM_U8_PREP: stem_u8.hip reads prep_input_u8's space-to-depth plane again instead of the raw image
    bool stem_u8_wg8 = false;                    // ..._STEM_U8_WG8: stem_u8.hip as ONE 8-wave workgroup per CU (8 x 32 conv tiles) instead of two 4-wave ones
    int stem_u8_seg = 0;                         // ..._STEM_U8_SEG = T: stem_u8.hip walks segments of T tiles (4 T - 1 pooled rows); 0 = its own choice, 1 = independent tiles
};
const Env& env();
void reload_env();

// ---- 16-bit float conversions (host + device) ------------------------------------------------
__host__ __device__ inline uint16_t f32_to_bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                           // RNE
    return (uint16_t)(u >> 16);
}
__host__ __device__ inline float bf16_bits_to_f32(uint16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__host__ __device__ inline uint16_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;  // RNE
    uint16_t b;
    memcpy(&b, &h, 2);
    return b;
}
__host__ __device__ inline float f16_bits_to_f32(uint16_t b) {
    _Float16 h;
    memcpy(&h, &b, 2);
    return (float)h;
}

// Element-type policies for the 16-bit activation/weight formats.
struct BF16 {
    typedef bf16x8_t frag_t;
    static constexpr int kDtype = DIR_BF16;
    __host__ __device__ static inline float to_f32(uint16_t b) { return bf16_bits_to_f32(b); }
    __host__ __device__ static inline uint16_t from_f32(float f) { return f32_to_bf16_bits(f); }
    __device__ static inline f32x16_t mfma32(frag_t a, frag_t b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    // two packed values <-> floats in one or two VALU ops (v_cvt_pk_bf16_f32 rounds to nearest even)
    __device__ static inline void unpack(uint32_t w, float& lo, float& hi) {
        lo = __builtin_bit_cast(float, w << 16);
        hi = __builtin_bit_cast(float, w & 0xffff0000u);
    }
    __device__ static inline uint32_t pack(float lo, float hi) {
        const f32x2_t v = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
    }
};
struct FP16 {
    typedef f16x8_t frag_t;
    static constexpr int kDtype = DIR_FP16;
    __host__ __device__ static inline float to_f32(uint16_t b) { return f16_bits_to_f32(b); }
    __host__ __device__ static inline uint16_t from_f32(float f) { return f32_to_f16_bits(f); }
    __device__ static inline f32x16_t mfma32(frag_t a, frag_t b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    __device__ static inline void unpack(uint32_t w, float& lo, float& hi) {
        const f32x2_t v = __builtin_convertvector(__builtin_bit_cast(f16x2_t, w), f32x2_t);
        lo = v[0];
        hi = v[1];
    }
    __device__ static inline uint32_t pack(float lo, float hi) {
        const f32x2_t v = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
    }
};

// Two packed 16-bit values <-> floats.
template <class DT>
__device__ inline void unpack2(uint32_t w, float& lo, float& hi) {
    lo = DT::to_f32((uint16_t)(w & 0xffffu));
    hi = DT::to_f32((uint16_t)(w >> 16));
}
template <class DT>
__device__ inline uint32_t pack2(float lo, float hi) {
    return (uint32_t)DT::from_f32(lo) | ((uint32_t)DT::from_f32(hi) << 16);
}

__device__ inline u32x4_t gload16(const void* p) {
    return *(const DIR_GLOBAL u32x4_t*)p;
}
__device__ inline void gstore16(void* p, u32x4_t v) {
    *(DIR_GLOBAL u32x4_t*)p = v;
}

// ---- fp16 overflow tripwire -----------------------------------------------------------------------
// The reference computes in fp32 and cannot overflow (dirtorch/nets/backbones/resnet.py:67-87); fp16 storage
// saturates at 65504.  Overflow can only happen where an fp32 value is packed for a store, and every packed
// value is stored, so watching the stores is a complete detector - a later ReLU (a hardware max, which drops a
// NaN operand) or a fused consumer cannot hide the event.  Each lane keeps the packed unsigned max of the
// |bit patterns| it stored (v_and + v_pk_max_u16 per two values); exponent all ones (inf / NaN) <=> pattern
// >= 0x7c00.  One atomicOr per offending lane at kernel end; the engine reads the word on request
// (dir_engine_overflow).  bf16 shares fp32's range: the tracker compiles to nothing.
typedef __attribute__((ext_vector_type(2))) uint16_t u16x2_t;
template <class DT>
struct Ovf {
    uint32_t m = 0;
    __device__ inline void see(uint32_t w) {
        if constexpr (DT::kDtype == DIR_FP16) {
            // (inline asm: as plain C the max chain is re-associated across the unrolled epilogue and the packed
            // words of several passes stay live - +18 VGPRs and a spill in the register-stationary kernels)
            uint32_t t;
            asm("v_and_b32 %1, 0x7fff7fff, %2\n\tv_pk_max_u16 %0, %0, %1" : "+v"(m), "=&v"(t) : "v"(w));
        }
    }
    __device__ inline void see(const u32x4_t& v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) see(v[e]);
    }
    __device__ inline void see(const u32x2_t& v) {
        see(v[0]);
        see(v[1]);
    }
    __device__ inline void flush(int* flag) const {
        if constexpr (DT::kDtype == DIR_FP16) {
            if (flag && ((m + 0x04000400u) & 0x80008000u)) atomicOr(flag, 1);
        }
    }
};

// ---- hand-off barrier of an LDS-DMA ring ------------------------------------------------------------------------------
// s_barrier with the instruction scheduler fenced on both sides.  hipcc (ROCm 7.2) treats neither
// `asm volatile("s_waitcnt vmcnt(N)" ::: "memory")` nor the raw s_barrier builtin as a fence for LDS READS when it
// schedules an unrolled K loop: in conv_patch3x3_kernel<64 channels> it had hoisted the first weight-fragment reads of
// stage t above the wait + barrier that make stage t visible (ds_read_b128 ... ; s_waitcnt vmcnt(2) ; s_barrier ;
// buffer_load ... lds ; v_mfma <those registers>).  Harmless while the stage - requested two steps earlier - has
// always landed by then, i.e. in every single-stream run; with forwards overlapping on several HIP streams about one
// launch in 2 400 read a half-landed stage (scripts/exp_stream_race_ops.py).  __builtin_amdgcn_sched_barrier(0) is the
// fence the scheduler honours (CDNA guide, section 5.4 rule 18); tests/test_isa_audit.py checks the emitted code of
// every ring kernel for LDS reads that cross a barrier into an MFMA.
__device__ __forceinline__ void ring_barrier() {
#ifndef DIR_EXP_NO_RING_FENCE   // experiment builds only (scripts/exp_abl.sh all DIR_EXP_NO_RING_FENCE 1): what the two fences cost
    __builtin_amdgcn_sched_barrier(0);
#endif
    __builtin_amdgcn_s_barrier();
#ifndef DIR_EXP_NO_RING_FENCE
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// Bijective XCD-aware remap of a 1-D block id: hardware places block b on XCD b % 8; give each
// XCD a contiguous run of logical tiles so neighbouring tiles (which share an operand panel) hit
// the same 4 MiB L2.  Placement only affects speed, never results.
__device__ inline int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// CU count of the current device, cached per device: hipGetDeviceProperties costs tens of microseconds, far too
// much for a launcher that runs 100+ times per batch-1 forward.
inline int cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int n = cached[dev & 63].load(std::memory_order_relaxed);
    if (n > 0) return n;
    hipDeviceProp_t prop;
    n = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    cached[dev & 63].store(n, std::memory_order_relaxed);
    return n;
}

// One-time opt-in of a kernel to more than 64 KiB of dynamic LDS, PER DEVICE (one engine per GPU may
// live in the same process) and safe against concurrent first launches from several host threads.
inline hipError_t ensure_dynamic_lds(const void* kern, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

}  // namespace dir
