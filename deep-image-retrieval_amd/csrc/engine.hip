// engine.hip — ResNet_RMAC descriptor extraction on one MI355X.
//
// What the reference does in Python with torch.nn modules
//   ResNet.forward        dirtorch/nets/backbones/resnet.py:157-174
//   Bottleneck.forward    dirtorch/nets/backbones/resnet.py:67-87   (BasicBlock :29-44)
//   ResNet_RMAC.forward   dirtorch/nets/rmac_resnet.py:39-69
// becomes a fixed launch sequence over caller-owned workspace:
//   prep (s2d) -> stem conv(+BN+ReLU) -> maxpool -> [bottleneck: 1x1 -> 3x3(s) -> 1x1 (+ds) +res +ReLU]*
//   -> global pool (GeM/max/avg, fp32) -> (L2) -> FC (fp32 MFMA) -> L2.
// Eval-mode BatchNorm (eps 1e-5) is folded into the conv weights and a per-channel fp32 bias at
// finalize(); activations stay NHWC 16-bit between kernels, every reduction is fp32.
#include "engine.h"

#include <math.h>
#include <algorithm>

namespace dir {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
const char* last_error() { return g_err.c_str(); }


static Env read_env() {
    Env e;
    auto on = [](const char* k) { return getenv(k) != nullptr; };
    const char* mode = getenv("DIRTORCH_AMD_C3C1");
    e.c3c1_off = mode && mode[0] == '0';
    e.c3c1_force = mode && mode[0] == 'f';
    e.no_ds_seam = on("DIRTORCH_AMD_NO_DS_SEAM");
    e.no_dual = on("DIRTORCH_AMD_NO_DUAL");
    e.rev_conv1 = on("DIRTORCH_AMD_REV_CONV1");
    e.rev_conv3 = on("DIRTORCH_AMD_REV_CONV3");
    e.unfused_stem = on("DIRTORCH_AMD_UNFUSED_STEM");
    e.stem_v1 = on("DIRTORCH_AMD_STEM_V1");
    e.no_patchlc = on("DIRTORCH_AMD_NO_PATCHLC");
    e.no_wreg = on("DIRTORCH_AMD_NO_WREG");
    e.no_wregd = on("DIRTORCH_AMD_NO_WREGD");
    e.no_smallmap = on("DIRTORCH_AMD_NO_SMALLMAP");
    e.no_c3c1lc = on("DIRTORCH_AMD_NO_C3C1LC");
    e.lc1x1 = on("DIRTORCH_AMD_LC1X1");
    e.x3_k2048 = on("DIRTORCH_AMD_X3_K2048");
    e.no_patchs2 = on("DIRTORCH_AMD_NO_PATCHS2");
    e.small_k2 = on("DIRTORCH_AMD_SMALL_K2");
    e.patchw_pack = on("DIRTORCH_AMD_PATCHW_PACK");
    e.no_patchw_pack = on("DIRTORCH_AMD_NO_PATCHW_PACK");
    e.no_patchw = on("DIRTORCH_AMD_NO_PATCHW");
    e.no_x3 = on("DIRTORCH_AMD_NO_X3");
    e.no_patchs = on("DIRTORCH_AMD_NO_PATCHS");
    e.no_patchw_lc = on("DIRTORCH_AMD_NO_PATCHW_LC");
    e.no_xcdmap = on("DIRTORCH_AMD_NO_XCDMAP");
    e.no_pair_patch = on("DIRTORCH_AMD_NO_PAIR_PATCH");
    e.pair_acts = on("DIRTORCH_AMD_PAIR_ACTS");
    if (const char* s = getenv("DIRTORCH_AMD_PAIR_STAGES")) e.pair_stages = atoi(s);
    e.sim_v1 = on("DIRTORCH_AMD_SIM_V1");
    e.sim_exact = on("DIRTORCH_AMD_SIM_EXACT");
    e.experiments = on("DIRTORCH_AMD_EXPERIMENTS");
    e.no_inplace = on("DIRTORCH_AMD_NO_INPLACE");
    e.no_stem_u8 = on("DIRTORCH_AMD_NO_STEM_U8");
    e.stem_u8_wg8 = on("DIRTORCH_AMD_STEM_U8_WG8");
    e.stem_u8_prep = on("DIRTORCH_AMD_STEM_U8_PREP");
    e.stem_pair_old = on("DIRTORCH_AMD_STEM_PAIR_OLD");
    if (const char* s = getenv("DIRTORCH_AMD_STEM_U8_SEG")) e.stem_u8_seg = atoi(s);
    return e;
}
// Two snapshots and an atomic index: dir_reload_env fills the idle one and publishes it with one store, so a launch that reads
// env() concurrently sees the old set or the new one, never a torn struct (round-5 advice).  (A reader that still holds a reference
// while a SECOND reload recycles its slot is the remaining window: reloads are a test / A-B tool, two per launch do not happen.)
static Env g_env[2];
static std::atomic<int> g_env_cur{-1};
static std::atomic_flag g_env_lock = ATOMIC_FLAG_INIT;
static void publish_env() {
    while (g_env_lock.test_and_set(std::memory_order_acquire)) {
    }
    const int cur = g_env_cur.load(std::memory_order_relaxed);
    const int nxt = cur < 0 ? 0 : 1 - cur;
    g_env[nxt] = read_env();
    g_env_cur.store(nxt, std::memory_order_release);
    g_env_lock.clear(std::memory_order_release);
}
const Env& env() {
    int cur = g_env_cur.load(std::memory_order_acquire);
    if (cur < 0) {
        static const bool once = (publish_env(), true);   // (thread-safe first use)
        (void)once;
        cur = g_env_cur.load(std::memory_order_acquire);
    }
    return g_env[cur];
}
void reload_env() { publish_env(); }   // host-driven (dir_reload_env)

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Deterministic noise in [-1, 1) for autotune inputs (no host RNG, no cuRAND analogue needed).
__global__ void fill_noise_kernel(uint16_t* p, long n, int dtype) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = (uint32_t)i * 2654435761u + 12345u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    const float f = (float)(h & 0xffff) / 32768.f - 1.f;
    p[i] = dtype == DIR_BF16 ? f32_to_bf16_bits(f) : f32_to_f16_bits(f);
}

}  // namespace dir

using namespace dir;

// ---- graph ------------------------------------------------------------------------------------
int dir_engine::build_graph() {
    convs.clear();
    blocks.clear();
    const int expansion = desc.bottleneck ? 4 : 1;
    auto add_conv = [&](const std::string& name, const std::string& wkey, const std::string& bn,
                        int cin, int cout, int k, int stride, int pad, bool relu) {
        ConvLayer L;
        L.name = name;
        L.wkey = wkey;
        L.bnprefix = bn;
        L.Cin = cin;
        L.Cout = cout;
        L.R = L.S = k;
        L.stride = stride;
        L.pad = pad;
        L.relu = relu;
        convs.push_back(L);
        return (int)convs.size() - 1;
    };
    // stem: 7x7 s2 p3, 3 -> 64 (resnet.py:115-118); executed as 4x4 s1 over the s2d image.
    {
        int i = add_conv("conv1", "conv1.weight", "bn1", 3, 64, 7, 2, 3, true);
        convs[i].stem = true;
    }
    int inplanes = 64;
    const int planes_of[4] = {64, 128, 256, 512};
    for (int s = 0; s < 4; ++s) {
        const int planes = planes_of[s];
        const int nblk = desc.layers[s];
        if (nblk <= 0) return fail(DIR_ERR_INVALID, "model desc: layers[] must be positive");
        for (int j = 0; j < nblk; ++j) {
            const int stride = (j == 0 && s > 0) ? 2 : 1;
            const std::string pre = "layer" + std::to_string(s + 1) + "." + std::to_string(j);
            BlockDef bd;
            bd.stride = stride;
            if (desc.bottleneck) {
                bd.conv1 = add_conv(pre + ".conv1", pre + ".conv1.weight", pre + ".bn1", inplanes,
                                    planes, 1, 1, 0, true);
                bd.conv2 = add_conv(pre + ".conv2", pre + ".conv2.weight", pre + ".bn2", planes,
                                    planes, 3, stride, 1, true);
                bd.conv3 = add_conv(pre + ".conv3", pre + ".conv3.weight", pre + ".bn3", planes,
                                    planes * 4, 1, 1, 0, true);  // ReLU after the residual add
            } else {
                bd.conv1 = add_conv(pre + ".conv1", pre + ".conv1.weight", pre + ".bn1", inplanes,
                                    planes, 3, stride, 1, true);
                bd.conv2 = add_conv(pre + ".conv2", pre + ".conv2.weight", pre + ".bn2", planes,
                                    planes, 3, 1, 1, true);  // ReLU after the residual add
            }
            if (j == 0 && (stride != 1 || inplanes != planes * expansion)) {
                // resnet.py:134-141: 1x1 conv (stride) + BN, no ReLU
                bd.down = add_conv(pre + ".downsample", pre + ".downsample.0.weight",
                                   pre + ".downsample.1", inplanes, planes * expansion, 1, stride, 0,
                                   false);
            }
            inplanes = planes * expansion;
            blocks.push_back(bd);
        }
    }
    feat_dim = inplanes;
    head_dim = feat_dim;
    x4_dim = 0;
    x4_block = conv1x5 = conv3c4 = -1;
    if (desc.head == DIR_HEAD_FPN || desc.head == DIR_HEAD_FPN0) {
        // rmac_resnet_fpn.py:24-32: dim1 = layer3 width, dim2 = layer4 width
        x4_dim = 256 * expansion;
        head_dim = x4_dim + feat_dim;
        x4_block = desc.layers[0] + desc.layers[1] + desc.layers[2] - 1;
        if (desc.head == DIR_HEAD_FPN) {
            // 1x1 conv commutes with the nearest upsample, so it runs on the small x5 grid
            conv1x5 = add_conv("conv1x5", "conv1x5.weight", "", feat_dim, x4_dim, 1, 1, 0, true);
            conv3c4 = add_conv("conv3c4", "conv3c4.weight", "", x4_dim, x4_dim, 3, 1, 1, true);
        }
    }
    return DIR_OK;
}

// ---- weights ------------------------------------------------------------------------------------
static const HostTensor* find(const std::map<std::string, HostTensor>& st, const std::string& k) {
    auto it = st.find(k);
    return it == st.end() ? nullptr : &it->second;
}

int dir_engine::finalize(int dt) {
    if (dt != DIR_BF16 && dt != DIR_FP16 && dt != DIR_F32 && dt != DIR_FP16P)
        return fail(DIR_ERR_INVALID, "finalize: bad dtype");
    DIR_HIP_CHECK(hipSetDevice(device));
    release();
    dtype = dt;
    auto to16 = [&](float f) { return dt == DIR_BF16 ? f32_to_bf16_bits(f) : f32_to_f16_bits(f); };
    // DIR_FP16P: the stem and the first pair_blocks residual blocks keep their weights as fp16 PAIRS (conv_pair.hip).
    // Default = all of layer1 (where tests/precision_decomposition.py puts the 16-bit error of a conditioned network);
    // DIRTORCH_AMD_PAIR_STAGES = 1..4 moves the boundary to the end of that stage.  Which weights: every conv of those
    // blocks with pair_acts (engine.h), else their 1x1s only.
    pair_blocks = 0;
    pair_acts = !desc.bottleneck || sw.pair_acts;
    std::vector<char> is_pair(convs.size(), 0);
    if (dt == DIR_FP16P) {
        const int stages = sw.pair_stages;
        if (stages < 1 || stages > 4) return fail(DIR_ERR_INVALID, "finalize: DIRTORCH_AMD_PAIR_STAGES must be 1..4");
        for (int s = 0; s < stages; ++s) pair_blocks += desc.layers[s];
        if (x4_block >= 0 && pair_blocks > x4_block)
            return fail(DIR_ERR_INVALID, "finalize: the FPN heads keep layer3's output as a single fp16 plane; "
                                         "DIRTORCH_AMD_PAIR_STAGES must stay below 3 for them");
        is_pair[0] = 1;
        for (int bi = 0; bi < pair_blocks; ++bi) {
            const BlockDef& bd = blocks[bi];
            for (int ci : {bd.conv1, bd.conv2, bd.conv3, bd.down})
                if (ci >= 0 && (pair_acts || (convs[ci].R == 1 && convs[ci].S == 1))) is_pair[ci] = 1;
        }
    }

    for (size_t li = 0; li < convs.size(); ++li) {
        ConvLayer& L = convs[li];
        const HostTensor* w = find(state, L.wkey);
        const HostTensor* g = find(state, L.bnprefix + ".weight");
        const HostTensor* bt = find(state, L.bnprefix + ".bias");
        const HostTensor* mu = find(state, L.bnprefix + ".running_mean");
        const HostTensor* var = find(state, L.bnprefix + ".running_var");
        if (!w) return fail(DIR_ERR_MISSING, "missing tensor " + L.wkey);
        const bool has_bn = !L.bnprefix.empty();
        if (has_bn && (!g || !bt || !mu || !var))
            return fail(DIR_ERR_MISSING, "missing BatchNorm tensors " + L.bnprefix + ".*");
        const int64_t want[4] = {L.Cout, L.Cin, L.R, L.S};
        if (w->shape.size() != 4 || !std::equal(want, want + 4, w->shape.begin()))
            return fail(DIR_ERR_INVALID, "bad shape for " + L.wkey);
        if (has_bn && ((int)g->data.size() != L.Cout || (int)bt->data.size() != L.Cout ||
                       (int)mu->data.size() != L.Cout || (int)var->data.size() != L.Cout))
            return fail(DIR_ERR_INVALID, "bad BatchNorm shape for " + L.bnprefix);

        std::vector<float> scale(L.Cout, 1.f), bias(L.Cout, 0.f);
        for (int o = 0; has_bn && o < L.Cout; ++o) {
            // BatchNorm2d eval: y = (x - mean) / sqrt(var + eps) * gamma + beta, eps = 1e-5
            const float inv = 1.0f / sqrtf(var->data[o] + 1e-5f);
            scale[o] = g->data[o] * inv;
            bias[o] = bt->data[o] - mu->data[o] * scale[o];
        }
        // packed in fp32 first ([Cout][R][S][Cin], BatchNorm scale folded in with one fp32 multiply); the
        // 16-bit formats round that, the strict path (DIR_F32, conv_f32.hip) uploads it as is
        std::vector<float> packed;
        if (L.stem) {
            // 7x7 s2 p3 over 3 channels == 4x4 s1 (pad 2 top/left) over the 2x2 space-to-depth
            // image with 12 (+4 zero) channels: tap r = 2R + dy - 1, s = 2S + dx - 1.
            packed.assign((size_t)L.Cout * 4 * 4 * 16, 0.f);
            for (int o = 0; o < L.Cout; ++o)
                for (int R = 0; R < 4; ++R)
                    for (int S = 0; S < 4; ++S)
                        for (int dy = 0; dy < 2; ++dy)
                            for (int dx = 0; dx < 2; ++dx) {
                                const int r = 2 * R + dy - 1, s = 2 * S + dx - 1;
                                if (r < 0 || s < 0 || r >= 7 || s >= 7) continue;
                                for (int c = 0; c < 3; ++c) {
                                    const float v =
                                        w->data[(((size_t)o * 3 + c) * 7 + r) * 7 + s] * scale[o];
                                    packed[(((size_t)o * 4 + R) * 4 + S) * 16 + (dy * 2 + dx) * 3 + c] = v;
                                }
                            }
        } else {
            packed.resize((size_t)L.Cout * L.R * L.S * L.Cin);
            for (int o = 0; o < L.Cout; ++o)
                for (int c = 0; c < L.Cin; ++c)
                    for (int r = 0; r < L.R; ++r)
                        for (int s = 0; s < L.S; ++s) {
                            const float v =
                                w->data[(((size_t)o * L.Cin + c) * L.R + r) * L.S + s] * scale[o];
                            packed[(((size_t)o * L.R + r) * L.S + s) * L.Cin + c] = v;
                        }
        }
        DIR_HIP_CHECK(hipMalloc((void**)&L.d_bias, L.Cout * 4));
        DIR_HIP_CHECK(hipMemcpy(L.d_bias, bias.data(), L.Cout * 4, hipMemcpyHostToDevice));
        L.tuned.clear();
        if (dt == DIR_F32) {
            DIR_HIP_CHECK(hipMalloc((void**)&L.d_wf, packed.size() * 4));
            DIR_HIP_CHECK(hipMemcpy(L.d_wf, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
            continue;
        }
        std::vector<uint16_t> packed16(packed.size());
        for (size_t i = 0; i < packed.size(); ++i) {
            packed16[i] = to16(packed[i]);
            // fp16 saturates at 65504: a folded weight beyond that would become inf here, on the host, where no
            // kernel's overflow word can see it (and inf * 0 = NaN is then flushed to 0 by the next fused ReLU)
            // (DIR_FP16P packs the same fp16 hi plane: hi = inf would make lo = fp16(w - inf) = -inf and hi + lo = NaN)
            if (dt != DIR_BF16 && (packed16[i] & 0x7c00u) == 0x7c00u && std::isfinite(packed[i]))
                return fail(DIR_ERR_RANGE, "finalize: a BatchNorm-folded weight of " + L.name + " (" +
                                               std::to_string(packed[i]) + ") exceeds the fp16 range; use DIR_BF16 or DIR_F32");
        }
        DIR_HIP_CHECK(hipMalloc((void**)&L.d_w, packed16.size() * 2));
        DIR_HIP_CHECK(hipMemcpy(L.d_w, packed16.data(), packed16.size() * 2, hipMemcpyHostToDevice));
        if (!L.stem && L.R == 3 && L.S == 3 && L.stride == 1 && L.pad == 1 && L.Cin % 32 == 0 && L.Cin >= 64 && L.Cout % 128 == 0) {
            // conv2 of layers 2-4: conv_patchw.hip's loaders copy their 24 KB weight stages as contiguous KBs from this copy
            DIR_HIP_CHECK(hipMalloc((void**)&L.d_w_pw, packed16.size() * 2));
            DIR_HIP_CHECK(conv_patch3x3w_pack(L.d_w, L.d_w_pw, L.Cout, L.Cin, nullptr));
        }
        if (!L.stem && L.R == 3 && L.S == 3 && L.stride == 2 && L.pad == 1 && L.Cin % 64 == 0 && L.Cout % 128 == 0) {
            // conv2 of a stage's first block: conv_patchs2.hip reads its weight fragments as contiguous KBs from this copy
            DIR_HIP_CHECK(hipMalloc((void**)&L.d_w_s2, packed16.size() * 2));
            DIR_HIP_CHECK(conv_patch3x3s2_pack(L.d_w, L.d_w_s2, L.Cout, L.Cin, nullptr));
        }
        if (L.stem && dt == DIR_FP16P) {   // (after the range check above: a weight the plain form refuses is reported as that layer's)
            const int rc = fold_stem_u8(L, w->data.data(), scale.data(), bias.data());
            if (rc != DIR_OK) return rc;
        }
        if (is_pair[li]) {   // lo plane: what the hi plane's rounding left over, itself rounded to fp16
            std::vector<uint16_t> lo16(packed.size());
            for (size_t i = 0; i < packed.size(); ++i) {
                lo16[i] = f32_to_f16_bits(packed[i] - f16_bits_to_f32(packed16[i]));
                if ((lo16[i] & 0x7c00u) == 0x7c00u && std::isfinite(packed[i]))   // (unreachable once hi is finite: |lo| <= ulp(hi) / 2)
                    return fail(DIR_ERR_RANGE, "finalize: the lo plane of a paired weight of " + L.name + " is not finite");
            }
            DIR_HIP_CHECK(hipMalloc((void**)&L.d_w_lo, lo16.size() * 2));
            DIR_HIP_CHECK(hipMemcpy(L.d_w_lo, lo16.data(), lo16.size() * 2, hipMemcpyHostToDevice));
            L.h_w_lo.swap(lo16);
        }
        L.h_w.swap(packed16);
        L.h_bias.swap(bias);
    }
    // First block of every stage (bottleneck nets): the residual is a 1x1 downsample conv of the block
    // input.  Its weights are appended to conv3's along K and the biases summed, so that
    // relu([W3 | Wds] . [t2 ; x_s] + b3 + bds) is ONE GEMM and the Cout-wide residual tensor is never
    // materialised: conv_c3c1's DS form (layer1: 64 + 64 channels, stride 1) or the two-source form of the
    // persistent 1x1 kernel (conv_persist.hip DUAL: layers 2-4, stride 2).
    for (const BlockDef& bd : blocks) {
        if (!desc.bottleneck || bd.down < 0 || bd.conv3 < 0 || dt == DIR_F32) continue;
        ConvLayer& c3 = convs[bd.conv3];
        const ConvLayer& ds = convs[bd.down];
        if (ds.R != 1 || ds.S != 1 || ds.pad != 0 || ds.Cout != c3.Cout || ds.Cin % 64 != 0 || c3.Cin % 64 != 0) continue;
        // paired head: conv_pair.hip's two-source form takes pixel-aligned sources of equal width only (layer1's first block)
        if (is_pair[bd.conv3] && (ds.stride != 1 || ds.Cin != c3.Cin || c3.Cout % 128 != 0)) continue;
        const int K3 = c3.Cin, Kd = ds.Cin, N = c3.Cout;
        std::vector<uint16_t> cat((size_t)N * (K3 + Kd));
        std::vector<float> bsum(N);
        for (int o = 0; o < N; ++o) {
            memcpy(&cat[(size_t)o * (K3 + Kd)], &c3.h_w[(size_t)o * K3], K3 * 2);
            memcpy(&cat[(size_t)o * (K3 + Kd) + K3], &ds.h_w[(size_t)o * Kd], Kd * 2);
            bsum[o] = c3.h_bias[o] + ds.h_bias[o];
        }
        DIR_HIP_CHECK(hipMalloc((void**)&c3.d_w_ds, cat.size() * 2));
        DIR_HIP_CHECK(hipMemcpy(c3.d_w_ds, cat.data(), cat.size() * 2, hipMemcpyHostToDevice));
        DIR_HIP_CHECK(hipMalloc((void**)&c3.d_bias_ds, N * 4));
        DIR_HIP_CHECK(hipMemcpy(c3.d_bias_ds, bsum.data(), N * 4, hipMemcpyHostToDevice));
        if (is_pair[bd.conv3]) {   // the lo planes, concatenated the same way
            for (int o = 0; o < N; ++o) {
                memcpy(&cat[(size_t)o * (K3 + Kd)], &c3.h_w_lo[(size_t)o * K3], K3 * 2);
                memcpy(&cat[(size_t)o * (K3 + Kd) + K3], &ds.h_w_lo[(size_t)o * Kd], Kd * 2);
            }
            DIR_HIP_CHECK(hipMalloc((void**)&c3.d_w_ds_lo, cat.size() * 2));
            DIR_HIP_CHECK(hipMemcpy(c3.d_w_ds_lo, cat.data(), cat.size() * 2, hipMemcpyHostToDevice));
        }
    }
    for (ConvLayer& L : convs) {
        std::vector<uint16_t>().swap(L.h_w_lo);
        std::vector<uint16_t>().swap(L.h_w);
        std::vector<float>().swap(L.h_bias);
    }

    const bool fpn = desc.head == DIR_HEAD_FPN || desc.head == DIR_HEAD_FPN0;
    if (desc.pooling == DIR_POOL_GEM && desc.head != DIR_HEAD_CLASSIFIER) {
        const char* keys[2] = {fpn ? "adpoolx5.p" : "adpool.p", "adpoolc4.p"};
        float* dst[2] = {&gem_p, &gem_p4};
        for (int i = 0; i < (fpn ? 2 : 1); ++i) {
            const HostTensor* p = find(state, keys[i]);
            if (!p || p->data.empty()) return fail(DIR_ERR_MISSING, std::string("missing tensor ") + keys[i]);
            *dst[i] = p->data[0];
            if (!(*dst[i] > 0.f)) return fail(DIR_ERR_INVALID, std::string(keys[i]) + " must be positive");
        }
    }
    if (!desc.without_fc) {
        const HostTensor* fw = find(state, "fc.weight");
        const HostTensor* fb = find(state, "fc.bias");
        if (!fw || !fb) return fail(DIR_ERR_MISSING, "missing tensor fc.weight / fc.bias");
        if (fw->shape.size() != 2 || fw->shape[0] != desc.out_dim || fw->shape[1] != head_dim ||
            (int)fb->data.size() != desc.out_dim)
            return fail(DIR_ERR_INVALID, "bad shape for fc.weight / fc.bias");
        DIR_HIP_CHECK(hipMalloc((void**)&d_fc_w, fw->data.size() * 4));
        DIR_HIP_CHECK(hipMemcpy(d_fc_w, fw->data.data(), fw->data.size() * 4, hipMemcpyHostToDevice));
        DIR_HIP_CHECK(hipMalloc((void**)&d_fc_b, fb->data.size() * 4));
        DIR_HIP_CHECK(hipMemcpy(d_fc_b, fb->data.data(), fb->data.size() * 4, hipMemcpyHostToDevice));
    }
    DIR_HIP_CHECK(hipMalloc((void**)&d_ovf, 256));
    DIR_HIP_CHECK(hipMemset(d_ovf, 0, 256));
    DIR_HIP_CHECK(hipDeviceSynchronize());
    finalized = true;
    return DIR_OK;
}

// ---- the stem for the RAW uint8 feed (stem_u8.hip) ------------------------------------------------------------------------
// ToTensor + Normalize (dirtorch/utils/transforms.py:617-623: x = (u / 255 - mean_c) / std_c) folded into conv1 + bn1
// (resnet.py:115-117).  The image plane holds u / 256 (exact in fp16), so
//   w'[o][tap][c] = w . bn_scale[o] . 256 / (255 std_c)                       an fp16 pair, packed like the s2d stem filter
//   b'[o]         = bn_bias[o] - sum_{all 147 taps} w . bn_scale[o] . mean_c / std_c
//   corr[rc][cc][o] = + sum_{taps OUTSIDE the image for border class (rc, cc)} w . bn_scale[o] . mean_c / std_c
// (a padded tap is zero in normalised space, i.e. it must NOT contribute its - mean / std; classes as stem_u8.hip's
// border_class: 0 all taps r = 0..6 inside, 1: r >= 3, 2: r >= 1, 3 / 4 / 5: r <= 5 / 4 / 3).  Sums in double.
int dir_engine::fold_stem_u8(ConvLayer& L, const float* w, const float* scale, const float* bias) {
    (void)L;
    std::vector<uint16_t> hi, lo;
    std::vector<float> b2, corr;
    const int rc = dir::fold_stem_u8(w, scale, bias, desc.mean, desc.std, hi, lo, b2, corr);
    // a preprocess whose 1 / (255 std) pushes a folded weight out of the fp16 range (or a std <= 0) only rules the uint8 stem out:
    // the tables stay null and dir_forward keeps the generic paired stem (normalisation in prep_input_pair) for that engine
    if (rc == DIR_ERR_RANGE || rc == DIR_ERR_INVALID) return DIR_OK;
    if (rc != DIR_OK) return rc;
    DIR_HIP_CHECK(hipMalloc((void**)&d_stem_u8_w, hi.size() * 2));
    DIR_HIP_CHECK(hipMemcpy(d_stem_u8_w, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
    DIR_HIP_CHECK(hipMalloc((void**)&d_stem_u8_w_lo, lo.size() * 2));
    DIR_HIP_CHECK(hipMemcpy(d_stem_u8_w_lo, lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
    DIR_HIP_CHECK(hipMalloc((void**)&d_stem_u8_bias, b2.size() * 4));
    DIR_HIP_CHECK(hipMemcpy(d_stem_u8_bias, b2.data(), b2.size() * 4, hipMemcpyHostToDevice));
    DIR_HIP_CHECK(hipMalloc((void**)&d_stem_u8_corr, corr.size() * 4));
    DIR_HIP_CHECK(hipMemcpy(d_stem_u8_corr, corr.data(), corr.size() * 4, hipMemcpyHostToDevice));
    return DIR_OK;
}

void dir_engine::release() {
    for (void** p : {(void**)&d_stem_u8_w, (void**)&d_stem_u8_w_lo, (void**)&d_stem_u8_bias, (void**)&d_stem_u8_corr}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    for (ConvLayer& L : convs) {
        if (L.d_w) (void)hipFree(L.d_w);
        if (L.d_wf) (void)hipFree(L.d_wf);
        L.d_wf = nullptr;
        if (L.d_w_lo) (void)hipFree(L.d_w_lo);
        L.d_w_lo = nullptr;
        if (L.d_w_s2) (void)hipFree(L.d_w_s2);
        if (L.d_w_pw) (void)hipFree(L.d_w_pw);
        L.d_w_s2 = L.d_w_pw = nullptr;
        if (L.d_bias) (void)hipFree(L.d_bias);
        if (L.d_w_ds) (void)hipFree(L.d_w_ds);
        if (L.d_w_ds_lo) (void)hipFree(L.d_w_ds_lo);
        L.d_w_ds_lo = nullptr;
        if (L.d_bias_ds) (void)hipFree(L.d_bias_ds);
        L.d_w = L.d_w_ds = nullptr;
        L.d_bias = L.d_bias_ds = nullptr;
    }
    if (d_fc_w) (void)hipFree(d_fc_w);
    if (d_fc_b) (void)hipFree(d_fc_b);
    d_fc_w = d_fc_b = nullptr;
    if (d_ovf) (void)hipFree(d_ovf);
    d_ovf = nullptr;
    finalized = false;
}

// ---- fp16 overflow word ------------------------------------------------------------------------------
// Reads (and clears) the word the kernels OR into whenever they store an fp16 inf / NaN.  Synchronises `stream`.
int dir_engine::overflow(hipStream_t stream, int* overflowed) {
    if (!finalized) return fail(DIR_ERR_STATE, "overflow query before finalize");
    int host = 0;
    DIR_HIP_CHECK(hipMemcpyAsync(&host, d_ovf, sizeof(int), hipMemcpyDeviceToHost, stream));
    DIR_HIP_CHECK(hipMemsetAsync(d_ovf, 0, sizeof(int), stream));
    DIR_HIP_CHECK(hipStreamSynchronize(stream));
    *overflowed = host != 0;
    return DIR_OK;
}

// ---- workspace plan ---------------------------------------------------------------------------
static inline int conv_out(int h, int k, int stride, int pad) { return (h + 2 * pad - k) / stride + 1; }

int dir_engine::plan(int B, int H, int W, Plan* p) const {
    if (B <= 0 || H <= 0 || W <= 0) return fail(DIR_ERR_INVALID, "forward: B, H, W must be positive");
    if (H < 7 || W < 7) return fail(DIR_ERR_INVALID, "forward: image smaller than the 7x7 stem");
    p->H2 = (H + 1) / 2;
    p->W2 = (W + 1) / 2;
    p->OH1 = conv_out(H, 7, 2, 3);
    p->OW1 = conv_out(W, 7, 2, 3);
    p->PH = conv_out(p->OH1, 3, 2, 1);
    p->PW = conv_out(p->OW1, 3, 2, 1);
    const size_t es = dtype == DIR_F32 ? 4 : 2;   // bytes per stored activation
    size_t io = 0, t1 = 0, t2 = 0, ds = 0, x4 = 0;
    int h = p->PH, w = p->PW;
    io = (size_t)B * h * w * 64 * es;
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
        const BlockDef& bd = blocks[bi];
        const int oh = conv_out(h, 3, bd.stride, 1), ow = conv_out(w, 3, bd.stride, 1);
        const ConvLayer& c1 = convs[bd.conv1];
        const ConvLayer& cl = convs[desc.bottleneck ? bd.conv3 : bd.conv2];
        if (desc.bottleneck) {
            t1 = std::max(t1, (size_t)B * h * w * c1.Cout * es);
            t2 = std::max(t2, (size_t)B * oh * ow * convs[bd.conv2].Cout * es);
        } else {
            t1 = std::max(t1, (size_t)B * oh * ow * c1.Cout * es);
        }
        if (bd.down >= 0) ds = std::max(ds, (size_t)B * oh * ow * convs[bd.down].Cout * es);
        io = std::max(io, (size_t)B * oh * ow * cl.Cout * es);
        h = oh;
        w = ow;
        if ((int)bi == x4_block) {  // FPN heads keep x4; the lateral path reuses t1 / t2 / a ping-pong buffer
            x4 = (size_t)B * h * w * x4_dim * es;
            t2 = std::max(t2, x4);
        }
    }
    if (conv1x5 >= 0) t1 = std::max(t1, (size_t)B * h * w * x4_dim * es);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes);
        return o;
    };
    p->s2d = take((size_t)B * p->H2 * p->W2 * 16 * es);
    p->stem = take((size_t)B * p->OH1 * p->OW1 * 64 * es);
    p->bufA = take(io);
    p->bufB = take(io);
    p->t1 = take(t1);
    p->t2 = take(t2 ? t2 : 256);
    p->ds = take(ds ? ds : 256);
    p->x4 = take(x4 ? x4 : 256);
    p->lo_s2d = p->lo_stem = p->lo_t1 = p->lo_t2 = p->lo_ds = 0;
    if (dtype == DIR_FP16P) {   // lo planes of the paired head: image, stem output [, t1 / t2 / downsample of its blocks]
        size_t pt1 = 0, pt2 = 0, pds = 0;
        int ph = p->PH, pw = p->PW;
        for (int bi = 0; pair_acts && bi < pair_blocks && bi < (int)blocks.size(); ++bi) {
            const BlockDef& bd = blocks[bi];
            const int oh = conv_out(ph, 3, bd.stride, 1), ow = conv_out(pw, 3, bd.stride, 1);
            const ConvLayer& c1 = convs[bd.conv1];
            if (desc.bottleneck) {
                pt1 = std::max(pt1, (size_t)B * ph * pw * c1.Cout * 2);
                pt2 = std::max(pt2, (size_t)B * oh * ow * convs[bd.conv2].Cout * 2);
            } else {
                pt1 = std::max(pt1, (size_t)B * oh * ow * c1.Cout * 2);
            }
            if (bd.down >= 0) pds = std::max(pds, (size_t)B * oh * ow * convs[bd.down].Cout * 2);
            ph = oh;
            pw = ow;
        }
        p->lo_s2d = take((size_t)B * p->H2 * p->W2 * 16 * 2);
        p->lo_stem = take((size_t)B * p->PH * p->PW * 64 * 2);
        p->lo_t1 = take(pt1 ? pt1 : 256);
        p->lo_t2 = take(pt2 ? pt2 : 256);
        p->lo_ds = take(pds ? pds : 256);
    }
    p->splitk = take(kSplitKMaxBytes);   // fp32 partial sums of split-K convs (small-M layers)
    p->pooled = take((size_t)B * head_dim * 4);
    p->fcout = take((size_t)B * std::max(desc.out_dim, head_dim) * 4);
    p->total = off;
    return DIR_OK;
}

// ---- profiling --------------------------------------------------------------------------------
int dir_engine::prof_begin(const std::string& name, const std::string& kernel, double flops,
                           double bytes, hipStream_t stream) {
    if (!profiling || prof_paused) return DIR_OK;
    if (prof_used == prof.size()) {
        ProfSlot s;
        DIR_HIP_CHECK(hipEventCreate(&s.start));
        DIR_HIP_CHECK(hipEventCreate(&s.stop));
        prof.push_back(s);
    }
    ProfSlot& s = prof[prof_used];
    s.name = name;
    s.kernel = kernel;
    s.flops = flops;
    s.bytes = bytes;
    DIR_HIP_CHECK(hipEventRecord(s.start, stream));
    return DIR_OK;
}
int dir_engine::prof_end(hipStream_t stream) {
    if (!profiling || prof_paused) return DIR_OK;
    DIR_HIP_CHECK(hipEventRecord(prof[prof_used].stop, stream));
    ++prof_used;
    return DIR_OK;
}

// ---- one convolution ----------------------------------------------------------------------------
int dir_engine::run_conv(ConvLayer& L, const uint16_t* x, const uint16_t* res, uint16_t* y, int B,
                         int H, int W, int OH, int OW, hipStream_t stream, bool rev_m) {
    // DIR_FP16P: a layer whose weights are a pair multiplies both planes (conv_pair.hip; single-plane input and output here)
    if (dtype == DIR_FP16P && L.d_w_lo && !L.stem)
        return run_conv_pair(L, x, nullptr, res, nullptr, y, nullptr, B, H, W, OH, OW, stream);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.rev_m = rev_m ? 1 : 0;
    a.x = x;
    a.w = L.d_w;
    a.w_s2 = L.d_w_s2;
    a.w_pw = env().no_patchw_pack ? nullptr : L.d_w_pw;
    a.bias = L.d_bias;
    a.res = res;
    a.y = y;
    a.B = B;
    a.H = H;
    a.W = W;
    a.OH = OH;
    a.OW = OW;
    a.Cout = L.Cout;
    if (L.stem) {  // (H, W) is the space-to-depth grid here
        a.Cin = 16;
        a.R = a.S = 4;
        a.stride = 1;
        a.pad = 2;
    } else {
        a.Cin = L.Cin;
        a.R = L.R;
        a.S = L.S;
        a.stride = L.stride;
        a.pad = L.pad;
    }
    a.relu = L.relu ? 1 : 0;
    a.ovf = d_ovf;
    a.partial = splitk_scratch;
    a.M = B * OH * OW;
    a.Ktot = a.R * a.S * a.Cin;
    a.T = a.Ktot / 64;
    const double macs = (double)a.M * L.Cout * (double)(L.R * L.S * L.Cin);  // true taps (stem: 147)
    const double bytes = 2.0 * ((double)B * H * W * a.Cin + (double)a.M * L.Cout * (res ? 2 : 1) +
                                (double)L.Cout * a.Ktot);

    int variant = -1;
    auto it = L.tuned.find(a.M);
    if (it != L.tuned.end()) variant = it->second;
    if (tuning && it == L.tuned.end()) {
        // time every admissible variant on the live input, keep the fastest
        hipEvent_t e0, e1;
        DIR_HIP_CHECK(hipEventCreate(&e0));
        DIR_HIP_CHECK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int v = 0; v < conv_variant_count(); ++v) {
            if (!conv_variant_admissible(v, a)) continue;
            a.ksplit = conv_splitk_factor(v, a);
            int rc = conv_launch(a, kdtype(), v, stream);  // warm-up (also sets func attributes)
            if (rc != DIR_OK) return rc;
            float ms = 1e30f;
            for (int round = 0; round < 2; ++round) {  // best of two timings of 3 launches
                DIR_HIP_CHECK(hipEventRecord(e0, stream));
                for (int rep = 0; rep < 3; ++rep) {
                    rc = conv_launch(a, kdtype(), v, stream);
                    if (rc != DIR_OK) return rc;
                }
                DIR_HIP_CHECK(hipEventRecord(e1, stream));
                DIR_HIP_CHECK(hipEventSynchronize(e1));
                float t = 0.f;
                DIR_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
                ms = t < ms ? t : ms;
            }
            if (ms < best) {
                best = ms;
                variant = v;
            }
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        L.tuned[a.M] = variant;
    }
    // a tuning table written for another architecture shares layer names (resnet18 / resnet101 both
    // have layer1.0.conv1): an entry that does not fit this layer's shape is dropped, not an error
    if (variant >= 0 && !conv_variant_admissible(variant, a)) {
        L.tuned.erase(a.M);
        variant = -1;
    }
    if (variant < 0) variant = conv_pick_variant(a);
    a.ksplit = conv_splitk_factor(variant, a);
    int rc = DIR_OK;
    if (profiling && !prof_paused)   // (the label strings are only built when somebody will read them)
        rc = prof_begin(L.name, std::string("conv_igemm<") + conv_variant(variant).name +
                                    (a.ksplit > 1 ? "/k" + std::to_string(a.ksplit) : "") + ">",
                        2.0 * macs, bytes, stream);
    if (rc != DIR_OK) return rc;
    rc = conv_launch(a, kdtype(), variant, stream);
    if (rc != DIR_OK) return rc;
    return prof_end(stream);
}

// ---- one convolution on fp16 pairs (DIR_FP16P head, conv_pair.hip) -------------------------------------------------
int dir_engine::run_conv_pair(ConvLayer& L, const uint16_t* x, const uint16_t* x_lo, const uint16_t* res,
                              const uint16_t* res_lo, uint16_t* y, uint16_t* y_lo, int B, int H, int W, int OH, int OW,
                              hipStream_t stream, const uint16_t* x2, const uint16_t* x2_lo) {
    if (!L.d_w_lo) return fail(DIR_ERR_STATE, "paired conv on a layer without a lo weight plane: " + L.name);
    if (x2 && !L.d_w_ds_lo) return fail(DIR_ERR_STATE, "two-source paired conv without concatenated weights: " + L.name);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.x_lo = x_lo;
    a.w = L.d_w;
    a.w_lo = L.d_w_lo;
    a.bias = L.d_bias;
    a.res = res;
    a.res_lo = res_lo;
    a.y = y;
    a.y_lo = y_lo;
    a.B = B;
    a.H = H;
    a.W = W;
    a.OH = OH;
    a.OW = OW;
    a.Cin = L.Cin;
    a.Cout = L.Cout;
    a.R = L.R;
    a.S = L.S;
    a.stride = L.stride;
    a.pad = L.pad;
    a.relu = L.relu ? 1 : 0;
    a.ovf = d_ovf;
    a.M = B * OH * OW;
    a.Ktot = a.R * a.S * a.Cin;
    if (x2) {   // conv3 + the block's stride-1 downsample: K = [t2 ; block input], weights / biases concatenated at finalize
        a.x2 = x2;
        a.x2_lo = x2_lo;
        a.Cin2 = a.Cin;
        a.w = L.d_w_ds;
        a.w_lo = L.d_w_ds_lo;
        a.bias = L.d_bias_ds;
        a.Ktot = 2 * a.Cin;
    }
    const double macs = (double)a.M * L.Cout * (double)a.Ktot;
    // every tensor a pair: twice the bytes of the 16-bit form (single-plane operands counted once)
    const double bytes = 2.0 * ((double)B * H * W * a.Cin * (x_lo ? 2 : 1) * (x2 ? 2 : 1) + (double)a.M * L.Cout * (y_lo ? 2 : 1) +
                                (double)a.M * L.Cout * (res ? (res_lo ? 2 : 1) : 0) + 2.0 * L.Cout * a.Ktot);
    int rc = DIR_OK;
    if (profiling && !prof_paused)
        rc = prof_begin(x2 ? L.name.substr(0, L.name.rfind('.')) + ".ds+conv3" : L.name,
                        std::string("conv_pair<") + conv_pair_variant_name(a) + ">", 2.0 * macs, bytes, stream);
    if (rc != DIR_OK) return rc;
    rc = conv_pair_launch(a, stream);
    if (rc != DIR_OK) return rc;
    return prof_end(stream);
}

// ---- DIR_FP16P: image -> stem -> the paired residual blocks ---------------------------------------------------------
// The reference's op sequence (ResNet.forward resnet.py:157-161, Bottleneck.forward :67-87 / BasicBlock :29-44), nothing
// fused across layers; every tensor two fp16 planes (hi in the ordinary workspace regions, lo in the lo_* regions).
int dir_engine::forward_pair_stem(const void* img, int B, int H, int W, int fmt, char* base, const Plan& p,
                                  hipStream_t stream) {
    uint16_t* s2d = (uint16_t*)(base + p.s2d);
    uint16_t* s2d_lo = (uint16_t*)(base + p.lo_s2d);
    int rc;
    if (img && fmt == DIR_IMG_U8_NHWC && d_stem_u8_w && !sw.no_stem_u8) {
        // the raw uint8 image is exact in ONE fp16 plane: Normalize folded into the filter pair and the bias, two MFMAs per
        // term, pooled in registers (stem_u8.hip)
        const bool raw = stem_pool_u8_raw_ok(img, B, H, W);   // the stem reads the image itself: no prep launch, no s2d plane
        if (!raw) {
            rc = prof_begin("prep_input", "prep_input_u8", 0, (double)B * H * W * 3 + (double)B * p.H2 * p.W2 * 32, stream);
            if (rc != DIR_OK) return rc;
            rc = prep_input_u8(img, s2d, B, H, W, stream);
            if (rc != DIR_OK) return rc;
            if ((rc = prof_end(stream)) != DIR_OK) return rc;
        }
        rc = prof_begin("conv1+maxpool", "stem_pool_u8", 2.0 * B * p.OH1 * p.OW1 * 64.0 * 147.0,
                        (raw ? (double)B * H * W * 3 : 2.0 * (double)B * p.H2 * p.W2 * 16) + 4.0 * ((double)B * p.PH * p.PW * 64 + 64 * 256),
                        stream);
        if (rc != DIR_OK) return rc;
        rc = stem_pool_u8_launch(raw ? img : nullptr, s2d, d_stem_u8_w, d_stem_u8_w_lo, d_stem_u8_bias, d_stem_u8_corr,
                                 (uint16_t*)(base + p.bufA), (uint16_t*)(base + p.lo_stem), B, H, W, stream, d_ovf, sw.stem_u8_seg);
        if (rc != DIR_OK) return rc;
        return prof_end(stream);
    }
    const bool raw_f32 = img && fmt == DIR_IMG_F32_NCHW && stem_pool_pair_raw_ok(img, B, H, W);   // the stem splits the fp32 image itself
    if (img && !raw_f32) {
        rc = prof_begin("prep_input", "prep_input_pair", 0, (double)B * H * W * 3 * (fmt == DIR_IMG_U8_NHWC ? 1 : 4) +
                        (double)B * p.H2 * p.W2 * 64, stream);
        if (rc != DIR_OK) return rc;
        rc = prep_input_pair(img, fmt, desc.mean, desc.std, s2d, s2d_lo, B, H, W, stream);
        if (rc != DIR_OK) return rc;
        if ((rc = prof_end(stream)) != DIR_OK) return rc;
    } else if (!img) {   // autotune: synthetic noise (the paired kernels themselves have nothing to tune)
        const long n = (long)B * p.H2 * p.W2 * 16;
        hipLaunchKernelGGL(fill_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, s2d, n, DIR_FP16);
        DIR_HIP_CHECK(hipGetLastError());
        DIR_HIP_CHECK(hipMemsetAsync(s2d_lo, 0, (size_t)n * 2, stream));
    }
    rc = prof_begin("conv1+maxpool", "stem_pool_pair", 2.0 * B * p.OH1 * p.OW1 * 64.0 * 147.0,
                    (raw_f32 ? (double)B * H * W * 12 : 4.0 * (double)B * p.H2 * p.W2 * 16) + 4.0 * ((double)B * p.PH * p.PW * 64 + 64 * 256), stream);
    if (rc != DIR_OK) return rc;
    if (raw_f32)
        rc = stem_pool_pair_walk_launch(nullptr, nullptr, convs[0].d_w, convs[0].d_w_lo, convs[0].d_bias, (uint16_t*)(base + p.bufA),
                                        (uint16_t*)(base + p.lo_stem), B, p.H2, p.W2, p.OH1, p.OW1, stream, d_ovf, img, H, W);
    else
        rc = stem_pool_pair_launch(s2d, s2d_lo, convs[0].d_w, convs[0].d_w_lo, convs[0].d_bias, (uint16_t*)(base + p.bufA),
                                   (uint16_t*)(base + p.lo_stem), B, p.H2, p.W2, p.OH1, p.OW1, stream, d_ovf);
    if (rc != DIR_OK) return rc;
    return prof_end(stream);
}

int dir_engine::forward_pair_head(const void* img, int B, int H, int W, int fmt, char* base, const Plan& p,
                                  hipStream_t stream, uint16_t** cur_out, int* h_out, int* w_out, size_t* next_block) {
    uint16_t* const pp[2] = {(uint16_t*)(base + p.bufA), (uint16_t*)(base + p.bufB)};
    uint16_t* stem_lo = (uint16_t*)(base + p.lo_stem);
    uint16_t* t1 = (uint16_t*)(base + p.t1);
    uint16_t* t2 = (uint16_t*)(base + p.t2);
    uint16_t* ds = (uint16_t*)(base + p.ds);
    uint16_t* t1_lo = (uint16_t*)(base + p.lo_t1);
    uint16_t* t2_lo = (uint16_t*)(base + p.lo_t2);
    uint16_t* ds_lo = (uint16_t*)(base + p.lo_ds);
    int rc = forward_pair_stem(img, B, H, W, fmt, base, p, stream);
    if (rc != DIR_OK) return rc;
    int cur = 0;

    // pair_acts form.  Which tensors are pairs (tests/precision_decomposition.py prices every storage point): the image,
    // the stem output and everything INSIDE a block (t1, t2, the downsample branch) - the block outputs, 4P wide and read three times
    // each, are single fp16 planes: their rounding costs 7e-6 of the 1e-4 budget and a third of the head's bytes.
    int h = p.PH, w = p.PW;
    const size_t nb = std::min((size_t)pair_blocks, blocks.size());
    for (size_t bi = 0; bi < nb; ++bi) {
        BlockDef& bd = blocks[bi];
        const int oh = conv_out(h, 3, bd.stride, 1), ow = conv_out(w, 3, bd.stride, 1);
        const int nxt = cur ^ 1;
        const uint16_t* x = pp[cur];
        const uint16_t* x_lo = bi == 0 ? stem_lo : nullptr;
        const uint16_t *resid = x, *resid_lo = x_lo;
        // first block of a stage: the downsample rides in conv3's GEMM when it is pixel-aligned with t2 (stride 1, same
        // width: layer1); otherwise it is its own paired conv
        const bool fuse_ds = bd.down >= 0 && desc.bottleneck && convs[bd.conv3].d_w_ds_lo != nullptr && x_lo != nullptr;
        if (bd.down >= 0 && !fuse_ds) {
            rc = run_conv_pair(convs[bd.down], x, x_lo, nullptr, nullptr, ds, ds_lo, B, h, w, oh, ow, stream);
            if (rc != DIR_OK) return rc;
            resid = ds;
            resid_lo = ds_lo;
        }
        if (desc.bottleneck) {
            rc = run_conv_pair(convs[bd.conv1], x, x_lo, nullptr, nullptr, t1, t1_lo, B, h, w, h, w, stream);
            if (rc != DIR_OK) return rc;
            rc = run_conv_pair(convs[bd.conv2], t1, t1_lo, nullptr, nullptr, t2, t2_lo, B, h, w, oh, ow, stream);
            if (rc != DIR_OK) return rc;
            if (fuse_ds)
                rc = run_conv_pair(convs[bd.conv3], t2, t2_lo, nullptr, nullptr, pp[nxt], nullptr, B, oh, ow, oh, ow, stream,
                                   x, x_lo);
            else
                rc = run_conv_pair(convs[bd.conv3], t2, t2_lo, resid, resid_lo, pp[nxt], nullptr, B, oh, ow, oh, ow, stream);
            if (rc != DIR_OK) return rc;
        } else {
            rc = run_conv_pair(convs[bd.conv1], x, x_lo, nullptr, nullptr, t1, t1_lo, B, h, w, oh, ow, stream);
            if (rc != DIR_OK) return rc;
            rc = run_conv_pair(convs[bd.conv2], t1, t1_lo, resid, resid_lo, pp[nxt], nullptr, B, oh, ow, oh, ow, stream);
            if (rc != DIR_OK) return rc;
        }
        cur = nxt;
        h = oh;
        w = ow;
    }
    *cur_out = pp[cur];
    *h_out = h;
    *w_out = w;
    *next_block = nb;
    return DIR_OK;
}

// ---- fused bottleneck seam ----------------------------------------------------------------------------
// The persistent seam kernel pays for loading both weight sets into registers once per workgroup: it wins from
// about 8 pixel tiles of 64 per workgroup (A/B at 1024^2: batch 1 = 1024 tiles: -2 %, batch 2: equal, batch 4:
// +4 %, batch 32: +6 %).
static constexpr long kSeamMinTiles = 2048;
static constexpr long kSeam3MinTiles = 768;   // conv_seam3.hip: three 64-pixel tiles per CU

int dir_engine::run_seam(ConvLayer& c3, ConvLayer& c1, const uint16_t* t2, const uint16_t* res, uint16_t* y,
                         uint16_t* t1, int B, int H, int W, hipStream_t stream, int* used,
                         const uint16_t* block_in, const uint16_t* block_in_lo) {
    *used = 0;
    // DIRTORCH_AMD_C3C1: "0" = never (A/B and bisecting), "force" = whenever the shapes qualify, default =
    // when every persistent workgroup gets at least ~8 pixel tiles to amortise loading both weight sets
    if (sw.c3c1_off) return DIR_OK;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = t2;
    a.w = c3.d_w;
    a.w_lo = c3.d_w_lo;       // DIR_FP16P: the weights of layer1's 1x1s are pairs (conv_c3c1.hip WP3 / WP1)
    a.w2_lo = c1.d_w_lo;
    a.bias = c3.d_bias;
    a.res = res;
    if (block_in) {   // DS form: the residual is the downsample conv of the block input, folded into this GEMM
        if (!c3.d_w_ds || c3.Cin != 64) return DIR_OK;   // (the caller checks the downsample's own shape)
        a.w = c3.d_w_ds;
        a.w_lo = c3.d_w_ds_lo;
        a.bias = c3.d_bias_ds;
        a.res = nullptr;
        a.x2 = block_in;
        a.x2_lo = block_in_lo;
        a.Cin2 = 64;
        if ((c3.d_w_lo != nullptr) != (a.w_lo != nullptr)) return DIR_OK;   // (a paired conv3 without the concatenated lo plane)
    }
    a.y = y;
    a.B = B;
    a.H = a.OH = H;
    a.W = a.OW = W;
    a.Cin = c3.Cin;
    a.Cout = c3.Cout;
    a.R = c3.R;
    a.S = c3.S;
    a.stride = c3.stride;
    a.pad = c3.pad;
    a.relu = c3.relu ? 1 : 0;
    a.M = B * H * W;
    a.Ktot = a.Cin;
    a.w2 = c1.d_w;
    a.bias2 = c1.d_bias;
    a.y2 = t1;
    a.Cout2 = c1.Cout;
    a.relu2 = c1.relu ? 1 : 0;
    a.ovf = d_ovf;
    if (c1.R != 1 || c1.S != 1 || c1.stride != 1 || c1.pad != 0 || c1.Cin != c3.Cout || !conv_c3c1_admissible(a))
        return DIR_OK;
    // (the layer3 form streams its weights per tile anyway: it only needs a few tiles per persistent workgroup)
    if (!sw.c3c1_force && (a.M + 63) / 64 < (a.Cin == 256 ? kSeam3MinTiles : kSeamMinTiles)) return DIR_OK;
    if (a.Cin == 256 && !sw.experiments) return DIR_OK;   // conv_seam3.hip: experiments builds only (it loses: profiles/r04_seam3_ablation.txt)
    const double macs = (double)a.M * ((double)c3.Cout * (c3.Cin + a.Cin2) + (double)c1.Cout * c1.Cin);
    const double bytes = 2.0 * ((double)a.M * (c3.Cin + a.Cin2 * (block_in_lo ? 2 : 1) + (block_in ? 1.0 : 2.0) * c3.Cout + c1.Cout) +
                                (double)c3.Cout * (c3.Cin + a.Cin2) * (a.w_lo ? 2 : 1) + (double)c1.Cout * c1.Cin * (a.w2_lo ? 2 : 1));
    // profile row "layerS.J.c3c1": conv3 of block J + conv1 of block J+1
    int rc = DIR_OK;
    if (profiling && !prof_paused)
        rc = prof_begin(c3.name.substr(0, c3.name.rfind('.')) + (block_in ? ".ds+c3c1" : ".c3c1"),
                        (c3.Cin == 256 ? "conv_seam3<" : "conv_c3c1<") + std::to_string(c3.Cin) + (block_in ? ",ds" : "") +
                            (a.w_lo ? ",wp>" : ">"),
                        2.0 * macs, bytes, stream);
    if (rc != DIR_OK) return rc;
    hipError_t e = conv_c3c1_launch(a, kdtype(), stream);
    if (e != hipSuccess) return fail(DIR_ERR_HIP, std::string("conv_c3c1 launch: ") + hipGetErrorString(e));
    *used = 1;
    return prof_end(stream);
}

// ---- conv3 + downsample as one two-source GEMM ------------------------------------------------------------
int dir_engine::run_conv_dual(ConvLayer& c3, const ConvLayer& ds, const uint16_t* t2, const uint16_t* xin,
                              uint16_t* y, int B, int Hin, int Win, int OH, int OW, hipStream_t stream, int* used,
                              bool dry) {
    *used = 0;
    if (!c3.d_w_ds || sw.no_dual || c3.d_w_lo) return DIR_OK;   // (paired weights: conv_pair.hip has no strided two-source form)
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = t2;
    a.w = c3.d_w_ds;
    a.bias = c3.d_bias_ds;
    a.y = y;
    a.B = B;
    a.H = a.OH = OH;
    a.W = a.OW = OW;
    a.Cin = c3.Cin;
    a.Cout = c3.Cout;
    a.R = a.S = 1;
    a.stride = 1;
    a.relu = c3.relu ? 1 : 0;
    a.ovf = d_ovf;
    a.M = B * OH * OW;
    a.x2 = xin;
    a.Cin2 = ds.Cin;
    a.H2 = Hin;
    a.W2 = Win;
    a.stride2 = ds.stride;
    a.Ktot = a.Cin + a.Cin2;
    a.T = a.Ktot / 64;
    const int variant = conv_pick_dual_variant(a);
    if (variant < 0) return DIR_OK;
    *used = 1;
    if (dry) return DIR_OK;
    const double macs = (double)a.M * c3.Cout * (double)a.Ktot;
    const double bytes = 2.0 * ((double)a.M * (c3.Cin + ds.Cin + c3.Cout) + (double)c3.Cout * a.Ktot);
    int rc = DIR_OK;
    if (profiling && !prof_paused)
        rc = prof_begin(c3.name.substr(0, c3.name.rfind('.')) + ".ds+conv3",
                        // ("conv_igemm<VARIANT>" is the family label of every entry of the variant table; the variant name says
                        // which kernel file runs it: 256x256_persist1x1_x3 = conv_persist.hip's deep-X ring, DUAL form)
                        std::string("conv_igemm<") + conv_variant(variant).name + "/dual>", 2.0 * macs, bytes, stream);
    if (rc != DIR_OK) return rc;
    rc = conv_launch(a, kdtype(), variant, stream);
    if (rc != DIR_OK) return rc;
    return prof_end(stream);
}

// ---- forward ------------------------------------------------------------------------------------
int dir_engine::forward(const void* img, int B, int H, int W, int fmt, float* desc_out,
                        void* feat_out, int* fh, int* fw, int* fc, void* ws, size_t ws_bytes,
                        hipStream_t stream) {
    if (!finalized) return fail(DIR_ERR_STATE, "forward before finalize");
    int cur_dev = -1;
    DIR_HIP_CHECK(hipGetDevice(&cur_dev));
    if (cur_dev != device)   // the weights live on `device`; launching elsewhere would fault
        return fail(DIR_ERR_STATE, "forward: the current HIP device (" + std::to_string(cur_dev) +
                                       ") is not the engine's device (" + std::to_string(device) + ")");
    Plan p;
    int rc = plan(B, H, W, &p);
    if (rc != DIR_OK) return rc;
    if (!ws || ws_bytes < p.total)
        return fail(DIR_ERR_WORKSPACE, "workspace too small: need " + std::to_string(p.total) +
                                           " bytes, got " + std::to_string(ws_bytes));
    if (((uintptr_t)ws & 255) != 0) return fail(DIR_ERR_INVALID, "workspace must be 256-byte aligned");
    char* base = (char*)ws;
    splitk_scratch = (float*)(base + p.splitk);
    if (dtype == DIR_F32) return forward_f32(img, B, H, W, fmt, desc_out, feat_out, fh, fw, fc, base, p, stream);
    uint16_t* s2d = (uint16_t*)(base + p.s2d);
    uint16_t* stem = (uint16_t*)(base + p.stem);
    uint16_t* cur = (uint16_t*)(base + p.bufA);
    uint16_t* nxt = nullptr;
    uint16_t* t1 = (uint16_t*)(base + p.t1);
    uint16_t* t2 = (uint16_t*)(base + p.t2);
    uint16_t* ds = (uint16_t*)(base + p.ds);
    float* pooled = (float*)(base + p.pooled);
    float* fcout = (float*)(base + p.fcout);

    const int kd = kdtype();
    int h = 0, w = 0;
    size_t first_block = 0;
    const uint16_t* cur_lo = nullptr;   // lo plane of the block input (DIR_FP16P: the stem's pooled output)
    if (dtype == DIR_FP16P && pair_acts) {
        // 1-2'. paired head: image, stem and the first pair_blocks residual blocks on fp16 pairs (conv_pair.hip)
        rc = forward_pair_head(img, B, H, W, fmt, base, p, stream, &cur, &h, &w, &first_block);
        if (rc != DIR_OK) return rc;
    } else if (dtype == DIR_FP16P) {
        // 1-2''. image and stem on pairs; the blocks run below on single fp16 planes, with paired 1x1 WEIGHTS in layer1
        rc = forward_pair_stem(img, B, H, W, fmt, base, p, stream);
        if (rc != DIR_OK) return rc;
        cur_lo = (const uint16_t*)(base + p.lo_stem);
        h = p.PH;
        w = p.PW;
    } else {
        // 1. image -> space-to-depth NHWC16
        if (img) {
            rc = prof_begin("prep_input", "prep_input", 0, (double)B * H * W * 3 * (fmt == DIR_IMG_U8_NHWC ? 1 : 4) +
                            (double)B * p.H2 * p.W2 * 32, stream);
            if (rc != DIR_OK) return rc;
            rc = prep_input(img, fmt, desc.mean, desc.std, s2d, B, H, W, kd, stream);
            if (rc != DIR_OK) return rc;
            if ((rc = prof_end(stream)) != DIR_OK) return rc;
        } else {  // autotune: synthetic noise straight into the s2d buffer
            const long n = (long)B * p.H2 * p.W2 * 16;
            hipLaunchKernelGGL(fill_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                               s2d, n, kd);
            DIR_HIP_CHECK(hipGetLastError());
        }
        // 2. stem + maxpool: one kernel (stem_pool.hip) unless DIRTORCH_AMD_UNFUSED_STEM=1
        if (sw.unfused_stem) {
            rc = run_conv(convs[0], s2d, nullptr, stem, B, p.H2, p.W2, p.OH1, p.OW1, stream);
            if (rc != DIR_OK) return rc;
            rc = prof_begin("maxpool", "maxpool_3x3s2", 0,
                            2.0 * ((double)B * p.OH1 * p.OW1 * 64 + (double)B * p.PH * p.PW * 64), stream);
            if (rc != DIR_OK) return rc;
            rc = maxpool_3x3s2(stem, cur, B, p.OH1, p.OW1, 64, kd, stream);
            if (rc != DIR_OK) return rc;
            if ((rc = prof_end(stream)) != DIR_OK) return rc;
        } else {
            rc = prof_begin("conv1+maxpool", "stem_pool", 2.0 * B * p.OH1 * p.OW1 * 64.0 * 147.0,
                            2.0 * ((double)B * p.H2 * p.W2 * 16 + (double)B * p.PH * p.PW * 64 + 64 * 256),
                            stream);
            if (rc != DIR_OK) return rc;
            rc = stem_pool_launch(s2d, convs[0].d_w, convs[0].d_bias, cur, B, p.H2, p.W2, p.OH1, p.OW1,
                                  kd, stream, d_ovf);
            if (rc != DIR_OK) return rc;
            if ((rc = prof_end(stream)) != DIR_OK) return rc;
        }
        h = p.PH;
        w = p.PW;
    }

    // 3. residual stages (two ping-pong buffers; the FPN heads park layer3's output in its own buffer)
    uint16_t* const pp[2] = {(uint16_t*)(base + p.bufA), (uint16_t*)(base + p.bufB)};
    uint16_t* x4 = nullptr;
    int h4 = 0, w4 = 0;
    bool t1_ready = false;
    for (size_t bi = first_block; bi < blocks.size(); ++bi) {
        BlockDef& bd = blocks[bi];
        const int oh = conv_out(h, 3, bd.stride, 1), ow = conv_out(w, 3, bd.stride, 1);
        const bool keep = (int)bi == x4_block;
        nxt = keep ? (uint16_t*)(base + p.x4) : (cur == pp[0] ? pp[1] : pp[0]);
        // The identity blocks of layers 3-4 write their output OVER their input (round 5; DIRTORCH_AMD_NO_INPLACE restores the
        // ping-pong).  Safe and bit-identical: conv3 reads every residual element it adds before it stores that element - conv_wreg
        // one strip ahead in registers, conv_persist / the tiled kernels the whole tile before the K loop, the split-K finalize
        // element by element - and no other workgroup touches it (conv3's INPUT is t2).  Found while pricing Infinity-Cache
        // residency (profiles/r05_mall_probe.txt: the cache itself buys nothing), kept for what it does in HBM: a
        // read-modify-write stream gets 5.6-5.9 TB/s where a copy gets 5.2-5.4 - layer3's conv3 110 -> 104 us, step 14.03 -> 13.96 ms
        // (A/B on one box) - and layer3's footprint in the workspace halves.
        if (!sw.no_inplace && desc.bottleneck && bd.down < 0 && !keep && !tuning && convs[bd.conv3].Cin >= 256) nxt = cur;
        const uint16_t* resid = cur;
        // layer1's first block: the downsample can ride in the seam kernel as extra K (conv_c3c1.hip, DS
        // form) - only if that kernel will actually run for this shape, decided before anything launches
        // (the next block's conv1 is 1x1 stride 1 even across a stage boundary - the stride sits in conv2 -
        // so the seam kernel also serves layer1 -> layer2; run_seam checks the widths it can hold)
        const bool seam_next = desc.bottleneck && bi + 1 < blocks.size() && !tuning;
        bool ds_in_seam = false;
        // (the DS form also needs the NEXT block's conv1 to be 64 wide - conv_c3c1_admissible: Cout2 == Cin; a
        // layer1 of a single block is followed by layer2's 128-wide conv1 and takes the two-source GEMM instead)
        if (bd.down >= 0 && seam_next && convs[bd.conv3].d_w_ds && convs[bd.conv3].Cin == 64 &&
            convs[bd.down].Cin == 64 && convs[bd.down].stride == 1 && convs[blocks[bi + 1].conv1].Cout == 64 &&
            convs[blocks[bi + 1].conv1].Cin == convs[bd.conv3].Cout) {
            // (oversized batches are left to conv_launch's own 2^31-byte error)
            ds_in_seam = !sw.c3c1_off && !sw.no_ds_seam && (sw.c3c1_force || ((long)B * oh * ow + 63) / 64 >= kSeamMinTiles) &&
                         (long)B * oh * ow * convs[bd.conv3].Cout < (1L << 30);
        }
        // the other stages' first blocks: conv3 + downsample as one two-source GEMM (conv_persist.hip, DUAL)
        int ds_dual = 0;
        if (bd.down >= 0 && !ds_in_seam && desc.bottleneck && !tuning && !cur_lo) {
            rc = run_conv_dual(convs[bd.conv3], convs[bd.down], t2, cur, nxt, B, h, w, oh, ow, stream, &ds_dual, true);
            if (rc != DIR_OK) return rc;
        }
        if (bd.down >= 0 && !ds_in_seam && !ds_dual) {
            rc = cur_lo ? run_conv_pair(convs[bd.down], cur, cur_lo, nullptr, nullptr, ds, nullptr, B, h, w, oh, ow, stream)
                        : run_conv(convs[bd.down], cur, nullptr, ds, B, h, w, oh, ow, stream);
            if (rc != DIR_OK) return rc;
            resid = ds;
        }
        if (desc.bottleneck) {
            if (!t1_ready) {   // (the previous block's fused seam kernel may have produced t1 already)
                rc = cur_lo ? run_conv_pair(convs[bd.conv1], cur, cur_lo, nullptr, nullptr, t1, nullptr, B, h, w, h, w, stream)
                            : run_conv(convs[bd.conv1], cur, nullptr, t1, B, h, w, h, w, stream, sw.rev_conv1);
                if (rc != DIR_OK) return rc;
            }
            t1_ready = false;
            rc = run_conv(convs[bd.conv2], t1, nullptr, t2, B, h, w, oh, ow, stream);
            if (rc != DIR_OK) return rc;
            int fused = 0;
            if (ds_dual) {
                rc = run_conv_dual(convs[bd.conv3], convs[bd.down], t2, cur, nxt, B, h, w, oh, ow, stream, &fused, false);
                if (rc != DIR_OK) return rc;
                if (!fused) return fail(DIR_ERR_STATE, "two-source conv declined a downsample it was promised");
                fused = 2;   // block output written; the next block's conv1 still has to run
            } else if (seam_next) {
                // conv3 + the next block's conv1 in one kernel: the block output is not re-read (conv_c3c1.hip)
                rc = run_seam(convs[bd.conv3], convs[blocks[bi + 1].conv1], t2, resid, nxt, t1, B, oh, ow, stream,
                              &fused, ds_in_seam ? cur : nullptr, ds_in_seam ? cur_lo : nullptr);
                if (rc != DIR_OK) return rc;
                if (ds_in_seam && !fused) return fail(DIR_ERR_STATE, "seam kernel declined a downsample it was promised");
            }
            if (fused == 1) {
                t1_ready = true;
            } else if (!fused) {
                rc = run_conv(convs[bd.conv3], t2, resid, nxt, B, oh, ow, oh, ow, stream, sw.rev_conv3);
                if (rc != DIR_OK) return rc;
            }
        } else {
            rc = run_conv(convs[bd.conv1], cur, nullptr, t1, B, h, w, oh, ow, stream);
            if (rc != DIR_OK) return rc;
            rc = run_conv(convs[bd.conv2], t1, resid, nxt, B, oh, ow, oh, ow, stream);
            if (rc != DIR_OK) return rc;
        }
        cur = nxt;
        cur_lo = nullptr;   // block outputs are single planes in every mode
        h = oh;
        w = ow;
        if (keep) {
            x4 = cur;
            h4 = h;
            w4 = w;
        }
    }
    if (feat_out) {
        DIR_HIP_CHECK(hipMemcpyAsync(feat_out, cur, (size_t)B * h * w * feat_dim * 2,
                                     hipMemcpyDeviceToDevice, stream));
        if (fh) *fh = h;
        if (fw) *fw = w;
        if (fc) *fc = feat_dim;
    }
    if (!desc_out) return DIR_OK;

    // 4. head (rmac_resnet.py:58-68; rmac_resnet_fpn.py:53-86; resnet.py:169-173)
    const bool fpn = x4 != nullptr;
    const bool classifier = desc.head == DIR_HEAD_CLASSIFIER;
    if (fpn) {
        const uint16_t* c4 = x4;
        if (conv1x5 >= 0) {
            uint16_t* sum = cur == pp[0] ? pp[1] : pp[0];
            rc = run_conv(convs[conv1x5], cur, nullptr, t1, B, h, w, h, w, stream);
            if (rc != DIR_OK) return rc;
            rc = prof_begin("x4+up(c5)", "upsample_add", 0,
                            2.0 * ((double)B * h4 * w4 * x4_dim * 2 + (double)B * h * w * x4_dim), stream);
            if (rc != DIR_OK) return rc;
            rc = upsample_add(x4, t1, sum, B, h4, w4, h, w, x4_dim, kd, stream, d_ovf);
            if (rc != DIR_OK) return rc;
            if ((rc = prof_end(stream)) != DIR_OK) return rc;
            rc = run_conv(convs[conv3c4], sum, nullptr, t2, B, h4, w4, h4, w4, stream);
            if (rc != DIR_OK) return rc;
            c4 = t2;
        }
        rc = prof_begin("adpoolc4", "global_pool", 0, (double)B * h4 * w4 * x4_dim * 2, stream);
        if (rc != DIR_OK) return rc;
        rc = global_pool(c4, pooled, head_dim, B, h4, w4, x4_dim, DIR_POOL_GEM, gem_p4, 1e-6f, 0.f,
                         kd, stream);
        if (rc != DIR_OK) return rc;
        if ((rc = prof_end(stream)) != DIR_OK) return rc;
    }
    rc = prof_begin(fpn ? "adpoolx5" : "adpool", "global_pool", 0,
                    (double)B * h * w * feat_dim * 2 + (double)B * feat_dim * 4, stream);
    if (rc != DIR_OK) return rc;
    // the FPN forward never applies center_bias (rmac_resnet_fpn.py:50-86), the classifier averages
    rc = global_pool(cur, pooled + (fpn ? x4_dim : 0), head_dim, B, h, w, feat_dim,
                     classifier ? DIR_POOL_AVG : desc.pooling, gem_p, 1e-6f,
                     (fpn || classifier) ? 0.f : desc.center_bias, kd, stream);
    if (rc != DIR_OK) return rc;
    if ((rc = prof_end(stream)) != DIR_OK) return rc;
    if (desc.norm_features && !classifier) {
        rc = l2norm_rows(pooled, B, head_dim, 1e-12f, stream);
        if (rc != DIR_OK) return rc;
    }
    const int D = desc.without_fc ? head_dim : desc.out_dim;
    if (!desc.without_fc) {
        rc = prof_begin("fc", "gemm_nt_f32", 2.0 * B * head_dim * (double)D,
                        4.0 * ((double)D * head_dim + (double)B * (head_dim + D)), stream);
        if (rc != DIR_OK) return rc;
        rc = gemm_nt_f32(d_fc_w, head_dim, pooled, head_dim, fcout, D, D, B, head_dim, nullptr,
                         d_fc_b, nullptr, stream, splitk_scratch, kSplitKMaxBytes);   // the convs are done with it
        if (rc != DIR_OK) return rc;
        if ((rc = prof_end(stream)) != DIR_OK) return rc;
    } else {
        fcout = pooled;
    }
    if (!classifier) {
        rc = l2norm_rows(fcout, B, D, 1e-12f, stream);
        if (rc != DIR_OK) return rc;
    }
    DIR_HIP_CHECK(hipMemcpyAsync(desc_out, fcout, (size_t)B * D * 4, hipMemcpyDeviceToDevice, stream));
    return DIR_OK;
}

// ---- the strict path: fp32 storage, fp32 matrix cores (conv_f32.hip) --------------------------------------------
// The reference's op sequence one to one (ResNet.forward resnet.py:157-174, Bottleneck.forward :67-87,
// ResNet_RMAC.forward rmac_resnet.py:39-69), eval-mode BatchNorm folded into fp32 weights, bias / residual / ReLU in
// the conv epilogue, nothing fused across layers.  Differences from the reference's fp32 CPU result are summation
// order only.
int dir_engine::run_conv_f32(ConvLayer& L, const float* x, const float* res, float* y, int B, int H, int W, int OH,
                             int OW, hipStream_t stream) {
    ConvF32Args a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.w = L.d_wf;
    a.bias = L.d_bias;
    a.res = res;
    a.y = y;
    a.B = B;
    a.H = H;
    a.W = W;
    a.OH = OH;
    a.OW = OW;
    a.Cout = L.Cout;
    if (L.stem) {  // (H, W) is the space-to-depth grid
        a.Cin = 16;
        a.R = a.S = 4;
        a.stride = 1;
        a.pad = 2;
    } else {
        a.Cin = L.Cin;
        a.R = L.R;
        a.S = L.S;
        a.stride = L.stride;
        a.pad = L.pad;
    }
    a.relu = L.relu ? 1 : 0;
    a.M = B * OH * OW;
    a.Ktot = a.R * a.S * a.Cin;
    const double macs = (double)a.M * L.Cout * (double)(L.R * L.S * L.Cin);
    const double bytes = 4.0 * ((double)B * H * W * a.Cin + (double)a.M * L.Cout * (res ? 2 : 1) + (double)L.Cout * a.Ktot);
    int rc = DIR_OK;
    if (profiling && !prof_paused)
        rc = prof_begin(L.name, std::string("conv_f32<") + (L.Cout <= 64 ? "128x64" : "128x128") + ">", 2.0 * macs, bytes,
                        stream);
    if (rc != DIR_OK) return rc;
    rc = conv_f32_launch(a, stream);
    if (rc != DIR_OK) return rc;
    return prof_end(stream);
}

int dir_engine::forward_f32(const void* img, int B, int H, int W, int fmt, float* desc_out, void* feat_out, int* fh,
                            int* fw, int* fc, char* base, const Plan& p, hipStream_t stream) {
    float* s2d = (float*)(base + p.s2d);
    float* stem = (float*)(base + p.stem);
    float* t1 = (float*)(base + p.t1);
    float* t2 = (float*)(base + p.t2);
    float* ds = (float*)(base + p.ds);
    float* pooled = (float*)(base + p.pooled);
    float* fcout = (float*)(base + p.fcout);
    float* const pp[2] = {(float*)(base + p.bufA), (float*)(base + p.bufB)};
    if (!img) return fail(DIR_ERR_INVALID, "autotune has nothing to choose in the fp32 path");
    int rc = prof_begin("prep_input", "prep_input_f32", 0,
                        (double)B * H * W * 3 * (fmt == DIR_IMG_U8_NHWC ? 1 : 4) + (double)B * p.H2 * p.W2 * 64, stream);
    if (rc != DIR_OK) return rc;
    rc = prep_input_f32(img, fmt, desc.mean, desc.std, s2d, B, H, W, stream);
    if (rc != DIR_OK) return rc;
    if ((rc = prof_end(stream)) != DIR_OK) return rc;
    rc = run_conv_f32(convs[0], s2d, nullptr, stem, B, p.H2, p.W2, p.OH1, p.OW1, stream);
    if (rc != DIR_OK) return rc;
    float* cur = pp[0];
    rc = prof_begin("maxpool", "maxpool_f32", 0, 4.0 * ((double)B * p.OH1 * p.OW1 * 64 + (double)B * p.PH * p.PW * 64),
                    stream);
    if (rc != DIR_OK) return rc;
    rc = maxpool_3x3s2_f32(stem, cur, B, p.OH1, p.OW1, 64, stream);
    if (rc != DIR_OK) return rc;
    if ((rc = prof_end(stream)) != DIR_OK) return rc;

    float* x4 = nullptr;
    int h = p.PH, w = p.PW, h4 = 0, w4 = 0;
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
        BlockDef& bd = blocks[bi];
        const int oh = conv_out(h, 3, bd.stride, 1), ow = conv_out(w, 3, bd.stride, 1);
        const bool keep = (int)bi == x4_block;
        float* nxt = keep ? (float*)(base + p.x4) : (cur == pp[0] ? pp[1] : pp[0]);
        const float* resid = cur;
        if (bd.down >= 0) {
            rc = run_conv_f32(convs[bd.down], cur, nullptr, ds, B, h, w, oh, ow, stream);
            if (rc != DIR_OK) return rc;
            resid = ds;
        }
        if (desc.bottleneck) {
            rc = run_conv_f32(convs[bd.conv1], cur, nullptr, t1, B, h, w, h, w, stream);
            if (rc != DIR_OK) return rc;
            rc = run_conv_f32(convs[bd.conv2], t1, nullptr, t2, B, h, w, oh, ow, stream);
            if (rc != DIR_OK) return rc;
            rc = run_conv_f32(convs[bd.conv3], t2, resid, nxt, B, oh, ow, oh, ow, stream);
            if (rc != DIR_OK) return rc;
        } else {
            rc = run_conv_f32(convs[bd.conv1], cur, nullptr, t1, B, h, w, oh, ow, stream);
            if (rc != DIR_OK) return rc;
            rc = run_conv_f32(convs[bd.conv2], t1, resid, nxt, B, oh, ow, oh, ow, stream);
            if (rc != DIR_OK) return rc;
        }
        cur = nxt;
        h = oh;
        w = ow;
        if (keep) {
            x4 = cur;
            h4 = h;
            w4 = w;
        }
    }
    if (feat_out) {
        DIR_HIP_CHECK(hipMemcpyAsync(feat_out, cur, (size_t)B * h * w * feat_dim * 4, hipMemcpyDeviceToDevice, stream));
        if (fh) *fh = h;
        if (fw) *fw = w;
        if (fc) *fc = feat_dim;
    }
    if (!desc_out) return DIR_OK;

    const bool fpn = x4 != nullptr;
    const bool classifier = desc.head == DIR_HEAD_CLASSIFIER;
    if (fpn) {
        const float* c4 = x4;
        if (conv1x5 >= 0) {
            float* sum = cur == pp[0] ? pp[1] : pp[0];
            rc = run_conv_f32(convs[conv1x5], cur, nullptr, t1, B, h, w, h, w, stream);
            if (rc != DIR_OK) return rc;
            rc = upsample_add_f32(x4, t1, sum, B, h4, w4, h, w, x4_dim, stream);
            if (rc != DIR_OK) return rc;
            rc = run_conv_f32(convs[conv3c4], sum, nullptr, t2, B, h4, w4, h4, w4, stream);
            if (rc != DIR_OK) return rc;
            c4 = t2;
        }
        rc = global_pool_f32(c4, pooled, head_dim, B, h4, w4, x4_dim, DIR_POOL_GEM, gem_p4, 1e-6f, 0.f, stream);
        if (rc != DIR_OK) return rc;
    }
    rc = prof_begin(fpn ? "adpoolx5" : "adpool", "global_pool_f32", 0,
                    (double)B * h * w * feat_dim * 4 + (double)B * feat_dim * 4, stream);
    if (rc != DIR_OK) return rc;
    rc = global_pool_f32(cur, pooled + (fpn ? x4_dim : 0), head_dim, B, h, w, feat_dim,
                         classifier ? DIR_POOL_AVG : desc.pooling, gem_p, 1e-6f,
                         (fpn || classifier) ? 0.f : desc.center_bias, stream);
    if (rc != DIR_OK) return rc;
    if ((rc = prof_end(stream)) != DIR_OK) return rc;
    if (desc.norm_features && !classifier) {
        rc = l2norm_rows(pooled, B, head_dim, 1e-12f, stream);
        if (rc != DIR_OK) return rc;
    }
    const int D = desc.without_fc ? head_dim : desc.out_dim;
    if (!desc.without_fc) {
        rc = prof_begin("fc", "gemm_nt_f32", 2.0 * B * head_dim * (double)D,
                        4.0 * ((double)D * head_dim + (double)B * (head_dim + D)), stream);
        if (rc != DIR_OK) return rc;
        rc = gemm_nt_f32(d_fc_w, head_dim, pooled, head_dim, fcout, D, D, B, head_dim, nullptr, d_fc_b, nullptr, stream,
                         splitk_scratch, kSplitKMaxBytes);
        if (rc != DIR_OK) return rc;
        if ((rc = prof_end(stream)) != DIR_OK) return rc;
    } else {
        fcout = pooled;
    }
    if (!classifier) {
        rc = l2norm_rows(fcout, B, D, 1e-12f, stream);
        if (rc != DIR_OK) return rc;
    }
    DIR_HIP_CHECK(hipMemcpyAsync(desc_out, fcout, (size_t)B * D * 4, hipMemcpyDeviceToDevice, stream));
    return DIR_OK;
}
