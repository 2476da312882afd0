/*
 * dir_engine.h — C ABI of the MI355X (gfx950) descriptor-extraction + ranking engine.
 *
 * The reference (naver/deep-image-retrieval, "dirtorch") has no FFI of its own: its seam is the
 * duck-typed Python object returned by dirtorch.nets.create_model (dirtorch/nets/__init__.py:24-64)
 * plus four free functions of dirtorch/utils/common.py.  Every entry point below names the
 * reference call it replaces.  The host side (Python, ctypes) lives in
 * deep-image-retrieval_amd/dirtorch_amd/ and mirrors the reference names one to one.
 *
 * Conventions
 *   - every function returns DIR_OK (0) or a negative dir_status; the message of the last error on
 *     the calling thread is returned by dir_last_error().  No C++ exception crosses this boundary.
 *   - all tensor pointers are DEVICE pointers owned by the caller unless the name says host_;
 *     `stream` is a hipStream_t passed as void* (NULL = the default stream).  Nothing here
 *     synchronises the device except dir_engine_finalize and the profiling getters.
 *   - activations are NHWC, 16-bit (bf16 or fp16; DIR_FP16P: pairs of fp16 planes in the stem and layer1) or fp32
 *     (DIR_F32, the strict path), chosen at finalize; accumulation is always fp32.
 *   - the engine owns only its packed weights; the caller owns images, descriptors and workspace.
 *   - one handle per device, ONE calling thread at a time (the reference calls net(x) from one thread,
 *     dirtorch/test_dir.py:67-81).  That thread may keep several dir_forward calls in flight on DIFFERENT streams of the
 *     device (the host mirror overlaps batch-1 forwards on four, dirtorch_amd/test_dir.py StreamPool): every per-forward
 *     mutable state is host-side and each call takes its own caller-owned workspace - one workspace per stream.  Two
 *     things are per handle, not per stream: the overflow word (join all streams before dir_engine_overflow) and the
 *     profile records (enable profiling with forwards on a single stream only - events of several streams interleave).
 */
#ifndef DIR_ENGINE_H
#define DIR_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum dir_status {
    DIR_OK = 0,
    DIR_ERR_INVALID = -1,     /* bad argument / unsupported configuration            */
    DIR_ERR_STATE = -2,       /* call out of order (e.g. forward before finalize)    */
    DIR_ERR_MISSING = -3,     /* a state-dict tensor was never supplied              */
    DIR_ERR_WORKSPACE = -4,   /* workspace too small                                 */
    DIR_ERR_HIP = -5,         /* a HIP runtime call failed                           */
    DIR_ERR_NOMEM = -6,
    DIR_ERR_RANGE = -7        /* a finite fp32 weight does not fit the chosen 16-bit format  */
} dir_status;

/* Storage format of activations and weights, chosen at dir_engine_finalize (the C ABI has no default; the Python
 * host mirror defaults to fp16, bench.py measures bf16 = BASELINE configs[1] and reports the other two beside it).
 * Accumulation is fp32 in all three. */
typedef enum dir_dtype {
    DIR_BF16 = 0,             /* bf16 storage, bf16 MFMA: fp32's exponent range, 8-bit mantissa               */
    DIR_FP16 = 1,             /* fp16 storage, fp16 MFMA: 11-bit mantissa, saturates at 65504 (dir_engine_overflow) */
    DIR_F32 = 2,              /* STRICT: fp32 storage and products on the fp32 matrix cores (conv_f32.hip) - the
                                 reference's own arithmetic up to summation order; ~1/8 of the 16-bit throughput */
    DIR_FP16P = 3             /* fp16 with a PAIRED head: where a conditioned network makes ~94 % of its 16-bit rounding
                                 error (the image, the stem, layer1) values are kept as pairs of fp16 planes (v ~ hi + lo,
                                 ~22 bits) and multiplied with two or three fp16 MFMAs per term - the image, the stem's
                                 weights and pooled output, and the weights of layer1's 1x1 convs (conv_pair.hip,
                                 conv_c3c1.hip WP); everything else is DIR_FP16.  Meets the 1e-4 cosine bar on
                                 BatchNorm-calibrated checkpoints (3.3e-5 ... 4.3e-5) at ~94 % of the fp16 throughput.
                                 Read at finalize: DIRTORCH_AMD_PAIR_ACTS=1 also pairs layer1's 3x3 weights and the
                                 tensors inside its blocks (1.5e-5 ... 1.7e-5, ~83 %; always on for BasicBlock nets),
                                 DIRTORCH_AMD_PAIR_STAGES=1..4 extends the paired region to later stages. */
} dir_dtype;

typedef enum dir_img_format {
    DIR_IMG_F32_NCHW = 0,     /* what the reference feeds net(x): normalised fp32 NCHW
                                 (dirtorch/utils/transforms.py:617-623 ToTensor+Normalize) */
    DIR_IMG_U8_NHWC = 1       /* raw uint8 HWC pixels; (x/255-mean)/std is fused on device  */
} dir_img_format;

typedef enum dir_pooling {    /* dirtorch/nets/rmac_resnet.py:24-31                         */
    DIR_POOL_GEM = 0,
    DIR_POOL_MAX = 1,
    DIR_POOL_AVG = 2
} dir_pooling;

typedef enum dir_head {       /* what follows the trunk                                     */
    DIR_HEAD_RMAC = 0,        /* ResNet_RMAC: pool -> (L2) -> FC -> L2 (rmac_resnet.py:39-69)       */
    DIR_HEAD_FPN = 1,         /* ResNet_RMAC_FPN mode 1: c4 = relu(conv3c4(x4 + up(relu(conv1x5(x5)))));
                                 GeM(c4) ++ GeM(x5) -> (L2) -> FC -> L2 (rmac_resnet_fpn.py:50-86)  */
    DIR_HEAD_FPN0 = 2,        /* mode 0 (resnet101_fpn0_rmac): GeM(x4) ++ GeM(x5), no lateral convs */
    DIR_HEAD_CLASSIFIER = 3   /* plain ResNet: avgpool -> FC, no L2 (backbones/resnet.py:169-174)   */
} dir_head;

/* Mirrors the keyword arguments of ResNet_RMAC.__init__ (dirtorch/nets/rmac_resnet.py:15-17)
 * and the layer counts of the resnet{18,50,101,152}_rmac factories (:74-88). */
typedef struct dir_model_desc {
    int   bottleneck;         /* 1 = Bottleneck blocks (R50/101/152), 0 = BasicBlock (R18)  */
    int   layers[4];          /* blocks per stage, e.g. {3,4,23,3}                          */
    int   out_dim;            /* FC output size (2048)                                      */
    int   norm_features;      /* L2 over channels before the FC                             */
    int   pooling;            /* dir_pooling                                                */
    int   without_fc;         /* skip the FC                                                */
    float center_bias;        /* >0: bilinear 4x4 centre mask before pooling (:52-56)       */
    float mean[3];            /* used by DIR_IMG_U8_NHWC only                               */
    float std[3];
    int   head;               /* dir_head; FPN heads need pooling == DIR_POOL_GEM and read the
                                 keys conv1x5.weight, conv3c4.weight, adpoolx5.p, adpoolc4.p   */
} dir_model_desc;

typedef struct dir_engine dir_engine;

const char* dir_last_error(void);
/* "dir_engine <version> gfx950" — lets the host check it loaded the library it expects. */
const char* dir_version(void);
/* The DIRTORCH_AMD_* A/B switches (kernel-selection toggles for bisecting; no reference counterpart - the reference reads no
 * environment on this path) are read from the environment ONCE, at the library's first use, and copied into an engine at
 * dir_engine_create.  A host that changes one of them afterwards (tests, A/B scripts) calls this to re-read them.  Two lifetimes:
 *   per ENGINE (copied at dir_engine_create; an existing engine keeps what it was created with):  _C3C1, _NO_DS_SEAM, _NO_DUAL,
 *       _REV_CONV1 / _REV_CONV3, _UNFUSED_STEM, _PAIR_ACTS, _PAIR_STAGES (the last two take effect at dir_engine_finalize),
 *       _NO_INPLACE, _NO_STEM_U8, _STEM_U8_SEG, _EXPERIMENTS (which kernels a forward may consider)
 *   per PROCESS (read from the current snapshot at every launch, by engines and by the per-op entry points alike):  the kernel
 *       pickers' _NO_PATCHLC / _NO_WREG / _NO_WREGD / _NO_SMALLMAP / _SMALL_K2 / _NO_C3C1LC / _LC1X1 / _X3_K2048 / _NO_PATCHS2 / _PATCHW_PACK / _NO_PATCHW_PACK / _NO_PATCHW / _NO_PATCHW_LC / _NO_X3 / _NO_PATCHS / _NO_PAIR_PATCH / _NO_XCDMAP,
 *       _STEM_V1, _STEM_PAIR_OLD, _STEM_U8_PREP, _STEM_U8_WG8, _SIM_V1, _SIM_EXACT
 * The re-read builds a new snapshot and publishes it with one atomic store (a launch sees the old set or the new one, never a
 * mixture); still, do not call it while another thread is launching if that thread's A/B comparison matters. */
int dir_reload_env(void);

/* ---- model life cycle: replaces nets.create_model + net.load_state_dict + net.cuda() ------- */
/* dirtorch/nets/__init__.py:24-64, dirtorch/test_dir.py:183-191 */
int dir_engine_create(const dir_model_desc* desc, int device, dir_engine** out);
int dir_engine_destroy(dir_engine* e);
/* One call per state-dict entry, reference key names (conv1.weight, bn1.running_mean,
 * layer3.7.conv2.weight, layer2.0.downsample.1.bias, adpool.p, fc.weight, fc.bias, …), host fp32,
 * PyTorch layouts (OIHW for convs).  num_batches_tracked entries are accepted and ignored. */
int dir_engine_set_tensor(dir_engine* e, const char* ref_key, const float* host_data,
                          const int64_t* shape, int ndim);
/* Folds eval-mode BatchNorm (eps 1e-5, dirtorch/nets/backbones/resnet.py:117) into the conv
 * weights, repacks OIHW -> [Cout][R][S][Cin] 16-bit, uploads.  DIR_ERR_MISSING names the key. */
int dir_engine_finalize(dir_engine* e, int dtype /* dir_dtype */);
int dir_engine_out_dim(const dir_engine* e, int* out_dim);

/* ---- forward: replaces ResNet_RMAC.forward (dirtorch/nets/rmac_resnet.py:39-69) ------------ */
int dir_workspace_bytes(const dir_engine* e, int B, int H, int W, size_t* bytes);
/* desc_out: B x D fp32, unit L2 norm.  (The reference's squeeze_ to [D] at B == 1 is a host-side
 * view, rmac_resnet.py:64.)  img: B x 3 x H x W fp32 or B x H x W x 3 uint8, per img_format. */
int dir_forward(dir_engine* e, const void* img, int B, int H, int W, int img_format,
                float* desc_out, void* workspace, size_t workspace_bytes, void* stream);
/* Same, but also returns the trunk feature map (NHWC, B x h x w x C, 16-bit or fp32 per the dtype) for block-level
 * parity tests against ResNet.forward (dirtorch/nets/backbones/resnet.py:157-174). */
int dir_forward_features(dir_engine* e, const void* img, int B, int H, int W, int img_format,
                         void* feat_out, int* h, int* w, int* c,
                         void* workspace, size_t workspace_bytes, void* stream);
/* Strict-path convolution (DIR_F32): y = act(conv(x, w) + bias (+ res)), everything fp32 NHWC / [Cout][R][S][Cin],
 * Cin % 4 == 0, Cout % 4 == 0; products on v_mfma_f32_32x32x2_f32 (an fmaf chain).  Replaces Conv2d + eval BatchNorm
 * (+ residual add) (+ ReLU) of dirtorch/nets/backbones/resnet.py:56-63,70-85 without any rounding of operands. */
int dir_conv_bn_act_f32(const float* x, const float* w, const float* bias, const float* res, float* y, int B, int H,
                        int W, int Cin, int Cout, int R, int S, int stride, int pad, int OH, int OW, int relu,
                        void* stream);
/* The paired-fp16 head of DIR_FP16P (csrc/conv_pair.hip).  Every operand is a PAIR of fp16 planes of the ordinary
 * layout, value = hi + lo with hi = fp16(v), lo = fp16(v - hi) (~22 significant bits); every product runs as three fp16
 * MFMAs into one fp32 accumulator (w_hi.x_hi + w_hi.x_lo + w_lo.x_hi).  Replaces, at ~fp32 accuracy,
 *   dir_conv_bn_act_pair  Conv2d + eval BatchNorm (+ residual add) (+ ReLU), dirtorch/nets/backbones/resnet.py:56-63,
 *                         70-85: x [B,H,W,Cin], w [Cout][R][S][Cin], res / y [B,OH,OW,Cout]; Cin % 32 == 0,
 *                         Cout % 64 == 0, R, S <= 4; x_lo / res_lo / y_lo may be NULL (single-plane operand / output)
 *   dir_conv_pair_dual    conv3 + bn3 + the block's STRIDE-1 downsample branch + add + ReLU of layer1's first bottleneck
 *                         as one GEMM over two pixel-aligned pair tensors of equal width (resnet.py:78-85 with :134-141):
 *                         y = act([w3 | wds] . [t2 ; x] + bias3 + bias_ds); t2, x [B,H,W,Cin], wcat [Cout][2 Cin],
 *                         Cout % 128 == 0; the 4P-wide downsample tensor is neither written nor read
 *   dir_prep_input_pair   ToTensor + Normalize (dirtorch/utils/transforms.py:617-623) -> space-to-depth pair
 *                         [B, ceil(H/2), ceil(W/2), 16] x 2
 *   dir_stem_pool_pair    conv 7x7 s2 + BN + ReLU + MaxPool 3x3 s2 (resnet.py:115-119) from that pair and the 4x4x16
 *                         packed filter pair to the pooled pair [B,PH,PW,64] x 2; the conv tile stays fp32 in LDS. */
int dir_conv_bn_act_pair(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                         const void* res_hi, const void* res_lo, void* y_hi, void* y_lo, int B, int H, int W, int Cin,
                         int Cout, int R, int S, int stride, int pad, int OH, int OW, int relu, void* stream);
int dir_conv_pair_dual(const void* t2_hi, const void* t2_lo, const void* x_hi, const void* x_lo, const void* wcat_hi,
                       const void* wcat_lo, const float* bias, void* y_hi, void* y_lo, int B, int H, int W, int Cin,
                       int Cout, int relu, void* stream);
int dir_prep_input_pair(const void* img, int img_format, const float* mean3, const float* std3, void* out_hi,
                        void* out_lo, int B, int H, int W, void* stream);
int dir_stem_pool_pair(const void* s2d_hi, const void* s2d_lo, const void* w_hi, const void* w_lo, const float* bias,
                       void* y_hi, void* y_lo, int B, int H2, int W2, int OH, int OW, void* stream);
/* DIR_FP16P on the RAW uint8 feed (stem_u8.hip; what dir_forward runs for DIR_IMG_U8_NHWC input): ToTensor + Normalize
 * (dirtorch/utils/transforms.py:617-623) + conv1 7x7 s2 + bn1 + ReLU + MaxPool 3x3 s2 (dirtorch/nets/backbones/resnet.py:115-119,
 * 158-161) from the uint8 NHWC image to the pooled fp16 pair [B,PH,PW,64] x 2.  The integers 0..255 are exact in one fp16 plane, so
 * the image needs no lo plane: the normalisation is folded into the filter PAIR (w . bn_scale . 256 / (255 std_c)), the bias and a
 * per-border-class bias table (zero padding happens AFTER Normalize in the reference), two MFMAs per term.  w_oihw: conv1.weight
 * [64,3,7,7] fp32 (host); bn_scale / bn_bias: bn1 in eval mode folded to y = conv * scale + bias (host, 64 each); s2d_ws: device
 * scratch of B * ceil(H/2) * ceil(W/2) * 32 bytes; seg_tiles: 0 = the kernel's own segment length, 1 = independent tiles.
 * Synchronises `stream` (parity entry point: the engine folds once at dir_engine_finalize). */
int dir_stem_pool_u8(const void* img_u8, const float* w_oihw, const float* bn_scale, const float* bn_bias, const float* mean3,
                     const float* std3, void* s2d_ws, void* y_hi, void* y_lo, int B, int H, int W, int seg_tiles, void* stream);
/* fp16 range check.  The reference computes in fp32 and cannot overflow (dirtorch/nets/backbones/resnet.py:67-87);
 * DIR_FP16 storage saturates at 65504.  Every kernel that packs fp32 sums into fp16 for a store ORs into an
 * engine-owned device word when it stores an inf / NaN - the first overflow of a forward is always such a store,
 * so a later ReLU (hardware max drops NaN operands) or a fused consumer cannot hide it.  This call synchronises
 * `stream`, returns the word in *overflowed (0 = clean) and clears it; the word is sticky across forwards until then.
 * Bit 0: an fp16 store overflowed.  Bit 1 (round 6): a kernel whose wave roles meet on LDS counters (conv_c3c1lc.hip, conv_small.hip)
 * gave up a bounded wait - its results are invalid; it cannot happen on a healthy device and is reported instead of hanging it.
 * Always 0 for DIR_BF16 (fp32's exponent range).  The host mirror raises FloatingPointError naming the
 * DIRTORCH_AMD_DTYPE switch (dirtorch_amd/test_dir.py _check_finite). */
int dir_engine_overflow(dir_engine* e, void* stream, int* overflowed);

/* Tile-variant selection for the implicit-GEMM convolutions: run every admissible variant on each
 * layer shape of a B x H x W forward and keep the fastest (synchronises).  Optional. */
int dir_engine_autotune(dir_engine* e, int B, int H, int W, void* workspace,
                        size_t workspace_bytes, void* stream);
/* Tuned choices as text, one "layer M variant-name" line each, so a later process (e.g. a
 * profiler run) can reuse them without re-timing.  export: *needed = bytes incl. NUL. */
int dir_engine_tuning_export(const dir_engine* e, char* buf, size_t cap, size_t* needed);
int dir_engine_tuning_import(dir_engine* e, const char* text);

/* ---- per-launch profile (HIP events on the launch stream) ---------------------------------- */
typedef struct dir_prof_record {
    char     name[48];        /* layer name, e.g. "layer3.4.conv2"                          */
    char     kernel[48];      /* kernel family + variant, e.g. "conv_igemm<128x128>"        */
    double   flops;           /* algorithmic FLOPs of this launch (2*MACs)                  */
    double   bytes;           /* algorithmic bytes (compulsory in + out + weights)          */
    float    ms;              /* event-measured duration                                    */
} dir_prof_record;
/* enabled: 0 off, 1 on, n > 1 on with n event pairs pre-created (keeps event creation out of
 * a timed region). */
int dir_engine_set_profiling(dir_engine* e, int enabled);
/* Suspend / resume recording without dropping what was recorded (sample every n-th step of a
 * timed loop so that the event records themselves stay out of most steps). */
int dir_engine_profile_pause(dir_engine* e, int paused);
/* Synchronises; copies up to `cap` records of the launches made since the last call. */
int dir_engine_get_profile(dir_engine* e, dir_prof_record* out, int cap, int* n);

/* ---- per-op entry points (op-level parity tests) ----------------------- */
/* K1/K2/K4-K7: conv + folded BN bias (+ residual) (+ ReLU); x NHWC [B,H,W,Cin] 16-bit,
 * w [Cout][R][S][Cin] 16-bit, bias fp32[Cout], res/y NHWC [B,OH,OW,Cout].
 * variant < 0: heuristic choice; otherwise index into dir_conv_variant_count(). Cin % 64 == 0,
 * or Cin == 16 with R == S == 4 (the space-to-depth stem). */
int dir_conv_variant_count(void);
int dir_conv_variant_name(int variant, char* buf, int cap);
int dir_conv_bn_act(const void* x, const void* w, const float* bias, const void* res, void* y,
                    int B, int H, int W, int Cin, int Cout, int R, int S, int stride,
                    int pad, int OH, int OW, int relu, int dtype, int variant, void* stream);
/* Which tile variant (and split-K factor) the engine picks for a layer shape that has not been autotuned.
 * Pure host logic - no GPU needed - exposed so that the choices distilled from the tuner stay pinned by
 * CPU tests (tests/test_capi_host.py). */
int dir_conv_heuristic(int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int OH,
                       int OW, int has_residual, char* name, int cap, int* ksplit);
/* Would dir_conv_bn_act accept `variant` for this shape?  *admissible = 1 / 0.  Pure host logic (the predicate conv_launch
 * itself applies), exposed so that the GPU parity matrix is generated from admissible (variant, shape) pairs only and the
 * tests' own mirror of the rule is pinned on the CPU (tests/test_capi_host.py). */
int dir_conv_variant_admissible(int variant, int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                int OH, int OW, int has_residual, int* admissible);
/* Same convolution with the K loop cut into `ksplit` slices that run as separate workgroups and meet in
 * an fp32 scratch buffer (ksplit * B*OH*OW * Cout floats; slices are added in a fixed order, then bias /
 * residual / ReLU) - what the engine does for layers with too few output tiles to fill 256 CUs (batch 1
 * at the deep stages).  ksplit < 0: the engine's own choice for that variant (reported in *ksplit_used);
 * variants without a split-K form accept only ksplit <= 1. */
int dir_conv_bn_act_splitk(const void* x, const void* w, const float* bias, const void* res, void* y,
                           int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                           int OH, int OW, int relu, int dtype, int variant, int ksplit, void* scratch,
                           size_t scratch_bytes, int* ksplit_used, void* stream);
/* The seam between two bottlenecks as ONE kernel (csrc/conv_c3c1.hip), planes P = 64 or 128:
 *   y  = act3(conv1x1(t2; w3 [4P][P]) + bias3 + res)      the conv3 + bn3 + add + ReLU that closes a block
 *   t1 = act1(conv1x1(y;  w1 [P2][4P]) + bias1)           the conv1 + bn1 + ReLU that opens the next one
 * (dirtorch/nets/backbones/resnet.py:78-85 then :70-72); P2 = P inside a stage, P2 = 128 after P = 64 for
 * the layer1 -> layer2 boundary.  t2 [B,H,W,P], res / y [B,H,W,4P], t1 [B,H,W,P2],
 * NHWC 16-bit; conv1 consumes the ROUNDED y, so the pair equals two dir_conv_bn_act calls up to fp32
 * summation order.  The engine uses it for the layer1 / layer2 seams when the map has >= 131072 pixels. */
int dir_conv_c3c1(const void* t2, const void* w3, const float* bias3, const void* res, void* y, const void* w1,
                  const float* bias1, void* t1, int B, int H, int W, int P, int P2, int relu3, int relu1, int dtype,
                  void* stream);
/* conv3 + bn3 + the block's downsample branch + add + ReLU of a stage's FIRST bottleneck as one GEMM over
 * two K sources (csrc/conv_persist.hip, DUAL form: the persistent deep-X ring with a second pixel source; dirtorch/nets/backbones/resnet.py:78-85 with :134-141):
 *   y[b,oh,ow,:] = act([w3 | wds] . [t2[b,oh,ow,:] ; x[b,oh*stride2,ow*stride2,:]] + bias),  bias = bias3 + bias_ds
 * t2 [B,OH,OW,Cin], x [B,H2,W2,Cin2] NHWC 16-bit, wcat [Cout][Cin + Cin2], Cout % 256 == 0, Cin, Cin2 % 64 == 0.
 * The Cout-wide residual tensor is neither written nor read. */
int dir_conv_dual(const void* t2, const void* x, const void* wcat, const float* bias, void* y, int B, int OH, int OW,
                  int Cin, int Cout, int Cin2, int H2, int W2, int stride2, int relu, int dtype, void* stream);
/* The same seam for the FIRST block of layer1 (planes 64), whose residual is the downsample branch
 * conv1x1(x; wds [256][64]) + bn of the 64-channel block input (resnet.py:134-141, 157-160): the
 * downsample is folded into the GEMM as 64 more K,
 *   y = act3([w3 | wds] . [t2 ; x] + (bias3 + bias_ds)),    wcat = [256][64 + 64], bias = the sum,
 * so the 256-wide residual tensor is never written or read.  t2, x [B,H,W,64]; y [B,H,W,256]; t1 [B,H,W,64]. */
int dir_conv_c3c1_ds(const void* t2, const void* x, const void* wcat, const float* bias, void* y, const void* w1,
                     const float* bias1, void* t1, int B, int H, int W, int relu3, int relu1, int dtype,
                     void* stream);
/* Both seams with PAIRED WEIGHTS (DIR_FP16P, fp16 only; csrc/conv_c3c1.hip WP3 / WP1): every weight matrix comes as two
 * fp16 planes, value = hi + lo with lo = fp16(w - hi) (~22 bits), and every product term costs two MFMAs instead of one -
 * no extra bytes on these HBM-bound kernels (measured: +6-9 % time, +35 % for the DS form with its third K block).  Activations are single fp16 planes, except the block input x of the DS form (the
 * stem's pooled output), which is a pair as well: [w3 | wds] . [t2 ; x_hi + x_lo] with the lo x lo term dropped.
 * P = 64 (layer1: dirtorch/nets/backbones/resnet.py:67-87, 134-141); P2 = 64 needs w1_lo, P2 = 128 (the layer1 ->
 * layer2 boundary, whose conv1 belongs to layer2) takes w1_lo = NULL for single-plane weights there. */
int dir_conv_c3c1_wpair(const void* t2, const void* w3, const void* w3_lo, const float* bias3, const void* res, void* y,
                        const void* w1, const void* w1_lo, const float* bias1, void* t1, int B, int H, int W, int P2,
                        int relu3, int relu1, void* stream);
int dir_conv_c3c1_ds_wpair(const void* t2, const void* x, const void* x_lo, const void* wcat, const void* wcat_lo,
                           const float* bias, void* y, const void* w1, const void* w1_lo, const float* bias1, void* t1,
                           int B, int H, int W, int relu3, int relu1, void* stream);
/* Slow, obviously-correct direct convolution with the same contract (device-side checker). */
int dir_conv_bn_act_naive(const void* x, const void* w, const float* bias, const void* res,
                          void* y, int B, int H, int W, int Cin, int Cout, int R, int S,
                          int stride, int pad, int OH, int OW, int relu, int dtype, void* stream);
/* Image -> space-to-depth stem input [B, ceil(H/2), ceil(W/2), 16] 16-bit (12 used channels). */
int dir_prep_input(const void* img, int img_format, const float* mean3, const float* std3,
                   void* out, int B, int H, int W, int dtype, void* stream);
/* K1+K2+K3 fused: the whole stem (conv 7x7 s2 + BN + ReLU + MaxPool 3x3 s2,
 * dirtorch/nets/backbones/resnet.py:115-119) from the space-to-depth image of dir_prep_input
 * [B,H2,W2,16] and the 4x4x16 packed stem filter to the pooled map [B,PH,PW,64]; OH x OW is the conv
 * output size the pool runs over.  What dir_forward uses. */
int dir_stem_pool(const void* s2d, const void* w, const float* bias, void* y, int B, int H2, int W2,
                  int OH, int OW, int dtype, void* stream);
/* The `Scale` test-time transform (dirtorch/utils/transforms.py:133-185 -> PIL
 * Image.resize(size, BILINEAR)) for uint8 NHWC RGB images already on the device: Pillow's
 * antialiased triangle filter with 22-bit fixed-point weights, horizontal then vertical pass, each
 * rounded to uint8 - bit-identical to Pillow (src/libImaging/Resample.c).  All B images share H x W.
 * workspace: dir_resize_workspace_bytes() bytes, 4-byte aligned; three async launches on `stream`. */
int dir_resize_workspace_bytes(int B, int H, int W, int OH, int OW, size_t* bytes);
int dir_resize_bilinear_u8(const void* src, void* dst, int B, int H, int W, int OH, int OW,
                           void* workspace, size_t workspace_bytes, void* stream);
/* K3: MaxPool2d(3, stride 2, pad 1) on NHWC (dirtorch/nets/backbones/resnet.py:119). */
int dir_maxpool_3x3s2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
/* K8: global pooling over H*W of NHWC x -> fp32 [B,C] (dirtorch/nets/layers/pooling.py:38-40);
 * pooling = dir_pooling, p = GeM exponent (runtime scalar), eps = 1e-6 clamp;
 * center_bias > 0 applies the bilinear mask of rmac_resnet.py:52-56 first. */
int dir_global_pool(const void* x, float* out, int B, int H, int W, int C, int pooling, float p,
                    float eps, float center_bias, int dtype, void* stream);
/* FPN lateral merge: y = x + nearest_upsample(low) with x,y NHWC [B,H,W,C], low [B,h,w,C], 16-bit;
 * source index min(floor(dst * in/out), in-1) as F.interpolate(mode='nearest', size=...) computes it
 * (dirtorch/nets/rmac_resnet_fpn.py:55-60).  C % 8 == 0. */
int dir_upsample_add(const void* x, const void* low, void* y, int B, int H, int W, int h, int w, int C,
                     int dtype, void* stream);
/* L2-normalise each row in place: x / max(||x||, eps) (F.normalize, rmac_resnet.py:7-9). */
int dir_l2norm_rows(float* x, int rows, int cols, float eps, void* stream);
/* K9/K11/K12: out[j][i] = alpha_i * (sum_k P[i][k] * (Q[j][k] - qsub[k])) + bias[i]
 *   P: [NP,K] fp32 (ldp), Q: [NQ,K] fp32 (ldq), out: [NQ,NP] fp32 (ldo); qsub/bias/alpha may be
 *   NULL.  Exact fp32 (v_mfma_f32_32x32x2_f32).  Any K >= 1 and any pitches ldp, ldq >= K (16-byte
 *   loads when K, ldp, ldq are multiples of 4 and the bases 16-byte aligned, an element-wise gather
 *   otherwise - e.g. --whitenv 50).  Used as
 *     FC          : P = fc.weight, Q = pooled features, bias = fc.bias   (rmac_resnet.py:66)
 *     PCA whiten  : P = components_[:v], Q = X, qsub = mean_, alpha = 1/(m*var^p) (common.py:221-232)
 *     similarity  : P = database descriptors, Q = queries -> scores[Q][N]       (common.py:30-38)
 *   Shapes with fewer than 128 output tiles (128 x 32..128) and K >= 512 - the FC of a batch, PCA whitening of a few
 *   hundred descriptors - run as up to 16 K slices whose fp32 partial sums (stream-ordered scratch, <= 64 MB) are
 *   added in slice order by a second kernel: same sums, another association, run-to-run identical.
 *   dir_gemm_splitk_factor (host-only) tells how many slices a shape gets. */
int dir_gemm_splitk_factor(int NP, int NQ, int K);
int dir_gemm_nt_f32(const float* P, int ldp, const float* Q, int ldq, float* out, int ldo,
                    int NP, int NQ, int K, const float* qsub, const float* bias,
                    const float* alpha, void* stream);
/* The three named uses of the GEMM above, as the reference's callers see them (all fp32, row-major,
 * contiguous; every output buffer is the caller's):
 *   dir_fc_l2          x[B,K] -> normalize(x W^T + b): the tail of ResNet_RMAC.forward
 *                      (dirtorch/nets/rmac_resnet.py:66-68); W is [D,K] (nn.Linear layout)
 *   dir_pca_whiten_l2  common.whiten_features (dirtorch/utils/common.py:221-239):
 *                      out[n][i] = scale[i] * sum_k components[i][k] * (X[n][k] - mean[k]), i < v, then
 *                      row L2 if l2norm != 0; scale[i] = 1 / (whitenm * explained_variance[i]^whitenp)
 *                      is computed by the caller (v floats, device), NULL = no rescaling
 *   dir_similarity     common.matmul(queries, database) (dirtorch/utils/common.py:30-38):
 *                      scores[q][n] = <queries[q], database[n]>, scores is [Q,N] with row stride N.
 *                      N >= 32768 with D % 32 == 0 (the 10^6-distractor protocol): every fp32 operand is split
 *                      into three bf16 planes and the six leading plane products run on the bf16 matrix cores
 *                      with fp32 accumulation (csrc/sim_split.hip) - products to 2^-23, measured error against
 *                      fp64 below that of the k-ordered fp32 chain, 1.8x its speed; takes ceil(Q/96) * D * 576
 *                      bytes of stream-ordered scratch (hipMallocAsync).  Smaller databases, other widths and
 *                      DIRTORCH_AMD_SIM_EXACT=1 in the environment: the exact k-ordered chain of dir_gemm_nt_f32.
 *   dir_similarity_unit  the same product for operands the caller KNOWS to lie in (-64, 64) - L2-normalised
 *                      descriptors, what common.matmul is called on in dirtorch/test_dir.py:150 - on the large-database
 *                      path: every operand as two fp16 planes of 2^10 x (~22 bits), three plane products instead of
 *                      six, scores rescaled by 2^-20: as accurate on unit vectors, 20+ % faster (the six-product form
 *                      is matrix-pipe-bound).  A value of magnitude >= 64 overflows its plane to inf and the scores
 *                      of its row come out NON-FINITE - never silently wrong.  Other sizes: identical to dir_similarity. */
int dir_fc_l2(const float* x, int B, int K, const float* W, const float* b, int D, float* out,
              void* stream);
int dir_pca_whiten_l2(const float* X, int N, int D, const float* mean, const float* components,
                      int v, const float* scale, int l2norm, float* out, void* stream);
/* dir_pca_whiten_l2 for operands the caller KNOWS to be bounded: |X - mean| < 64 and |components| < 64 - L2-normalised descriptors
 * and the orthonormal rows of a PCA, what common.whiten_features is called on in dirtorch/test_dir.py:136-138.  N >= 32768 with
 * D % 32 == 0 and D <= 6144 (the 10^6-distractor database of BASELINE configs[3]: 8.4 TFLOP): X minus the mean (subtracted in fp32,
 * before anything is rounded) and the components as two fp16 planes of 2^10 x each (~22 bits), three plane products on the fp16
 * matrix cores, fp32 accumulation, result x 2^-20 scale[j] (csrc/sim_split.hip whiten_split_kernel): within ~1e-6 (relative to the
 * row's largest entry) of the fp64 product, ~4x the fp32 MFMA chain.  A value outside the range overflows its plane and its row
 * comes out NON-FINITE, never silently wrong.  Other sizes, and DIRTORCH_AMD_SIM_EXACT=1: identical to dir_pca_whiten_l2.
 * Takes ceil(v/96) * D * 384 bytes of stream-ordered scratch (hipMallocAsync). */
int dir_pca_whiten_l2_unit(const float* X, int N, int D, const float* mean, const float* components,
                      int v, const float* scale, int l2norm, float* out, void* stream);
int dir_similarity(const float* queries, int Q, const float* database, int N, int D, float* scores,
                   void* stream);
int dir_similarity_unit(const float* queries, int Q, const float* database, int N, int D, float* scores,
                        void* stream);
/* K10: multi-scale pooling of S descriptor sets [S][N][D] -> [N][D] (common.py:41-55):
 * mode 0 = mean, 1 = signed-power ("gem") mean with exponent gemp; no final L2 (caller does it). */
int dir_multiscale_pool(const float* x, float* out, int S, int N, int D, int mode, float gemp,
                        void* stream);

/* N1: ranking without sorting, for ImageListRelevants.eval_query_AP
 * (dirtorch/datasets/generic.py:196-224).  For every query q and every probe p (a database index,
 * -1 = unused slot) counts[q][p] = number of database items that rank before it under
 * np.argsort(scores[q])[::-1]: score greater, or equal with a larger index.  probe_scores[q][p]
 * receives scores[q][probe].  scores: [Q][N] fp32 with row stride lds; probe_idx / counts /
 * probe_scores are [Q][P] (any P: 4096 probes per query per pass).  Sorted probes + one binary search per score + a
 * histogram (ranking.hip): each score is read once per pass, NaN scores rank before nothing, -0 == +0; probe_scores
 * doubles as the kernels' scratch until the call's last kernel (no workspace argument). */
int dir_rank_counts(const float* scores, int lds, int Q, int N, const int* probe_idx, int P,
                    int* counts, float* probe_scores, void* stream);
/* N1, second half: the APs themselves on the device - ImageListRelevants.eval_query_AP
 * (dirtorch/datasets/generic.py:196-224) + compute_average_precision (dirtorch/utils/evaluation.py:46-82)
 * from the outputs of dir_rank_counts.  For every (query q, mode m < modes) the caller lists, as
 * positions INTO row q of probe_idx, the positives pos_list[pos_off[q*modes+m] .. pos_off[q*modes+m+1])
 * and the junk junk_list[junk_off[..] ..) of that mode (duplicates removed; an image that is both is
 * junk - generic.py:192).  ap_out[q*modes+m] = the trapezoidal AP, summed in fp64 in the reference's
 * order, or -1 when the mode has no positive.  terms: fp64 scratch, one slot per pos_list entry. */
int dir_revisitop_ap(const int* probe_idx, int Q, int P, const int* counts, const float* probe_scores,
                     const int* pos_off, const int* pos_list, const int* junk_off, const int* junk_list,
                     int modes, double* terms, double* ap_out, void* stream);
/* N3: alpha query expansion / database augmentation, expand_descriptors of
 * dirtorch/test_dir.py:24-44.  out[i] = normalize(mean(descs[i], sim[i][j]^alpha * db[j] for the k
 * rows j of db most similar to descs[i])), sim = descs . db^T in fp32; self_set != 0 (db == descs,
 * m == n): a row is not its own neighbour (its self-similarity counts as 0, test_dir.py:33-34).
 * descs [n][D], db [m][D], out [n][D] fp32 contiguous; sim: caller's scratch of sim_bytes >= 4*m
 * (rows are processed in chunks of sim_bytes / (4*m)); 0 <= k <= m, alpha >= 0.  A row whose similarities are
 * not finite yields a NaN row (the reference propagates NaN the same way). */
int dir_expand_descriptors(const float* descs, int n, const float* db, int m, int D, int k, float alpha,
                           int self_set, float* out, float* sim, size_t sim_bytes, void* stream);

/* ---- the exchange step ------------------------------------------------------------
 * All-gather of per-GPU descriptor blocks over RCCL / xGMI for ONE process driving several GPUs - the
 * shape of the reference's nn.DataParallel use (dirtorch/utils/common.py:150-175).  One process per
 * GPU reaches the same collective through torch.distributed (dirtorch_amd/distributed.py).  librccl is
 * bound at first use (dlopen): without it these calls fail with DIR_ERR_STATE, nothing else changes.
 *   dir_comm_init_all   ncclCommInitAll over devices[0..ndev) (NULL = 0..ndev-1)
 *   dir_allgather_desc  rank i contributes send[i] = [rows][D] fp32 on device i and receives
 *                       recv[i] = [ndev*rows][D] (rank order = shard order; equal `rows` per rank - the
 *                       caller pads the last shard and trims after, as distributed.py does);
 *                       streams[i] = the stream of device i to enqueue on (NULL array = default streams) */
typedef struct dir_comm dir_comm;
int dir_comm_init_all(int ndev, const int* devices, dir_comm** out);
int dir_comm_size(const dir_comm* c, int* ndev);
int dir_comm_destroy(dir_comm* c);
int dir_allgather_desc(dir_comm* c, const float* const* send, float* const* recv, size_t rows, int D,
                       void* const* streams);

#ifdef __cplusplus
}
#endif
#endif /* DIR_ENGINE_H */
