// c_api.hip — the extern "C" surface declared in include/dir_engine.h.
// Nothing above this file knows about Python or torch; nothing below it throws.
#include <new>

#include "engine.h"
#include "pointwise.h"

namespace dir {
const char* last_error();
}
using namespace dir;

// databases at least this long take the split-bf16 similarity kernel (sim_split.hip)
static constexpr int kSimSplitMinRows = 32768;

#define DIR_TRY try {
#define DIR_CATCH                                                   \
    }                                                               \
    catch (const std::bad_alloc&) {                                 \
        return fail(DIR_ERR_NOMEM, "out of host memory");           \
    }                                                               \
    catch (const std::exception& ex) {                              \
        return fail(DIR_ERR_INVALID, std::string("exception: ") + ex.what()); \
    }                                                               \
    catch (...) {                                                   \
        return fail(DIR_ERR_INVALID, "unknown exception");          \
    }

extern "C" {

const char* dir_last_error(void) { return last_error(); }
const char* dir_version(void) { return "dir_engine 0.1 gfx950"; }

int dir_reload_env(void) {
    dir::reload_env();
    return DIR_OK;
}

int dir_engine_create(const dir_model_desc* desc, int device, dir_engine** out) {
    DIR_TRY
    if (!desc || !out) return fail(DIR_ERR_INVALID, "create: null argument");
    if (desc->pooling < DIR_POOL_GEM || desc->pooling > DIR_POOL_AVG)
        return fail(DIR_ERR_INVALID, "create: bad pooling mode");  // ValueError(pooling), rmac_resnet.py:31
    if (!desc->without_fc && desc->out_dim <= 0) return fail(DIR_ERR_INVALID, "create: out_dim <= 0");
    if (desc->head < DIR_HEAD_RMAC || desc->head > DIR_HEAD_CLASSIFIER)
        return fail(DIR_ERR_INVALID, "create: bad head");
    if ((desc->head == DIR_HEAD_FPN || desc->head == DIR_HEAD_FPN0) && desc->pooling != DIR_POOL_GEM)
        // the reference only builds adpoolx5/adpoolc4 for 'gem' (rmac_resnet_fpn.py:39-45)
        return fail(DIR_ERR_INVALID, "create: the FPN heads exist for GeM pooling only");
    if (desc->head == DIR_HEAD_CLASSIFIER && desc->without_fc)
        return fail(DIR_ERR_INVALID, "create: the classifier head needs its FC");
    int ndev = 0;
    DIR_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(DIR_ERR_INVALID, "create: no such device");
    dir_engine* e = new dir_engine();
    e->desc = *desc;
    e->device = device;
    e->sw = dir::env();   // the A/B switches as they stand now; forward() never reads the environment
    int rc = e->build_graph();
    if (rc != DIR_OK) {
        delete e;
        return rc;
    }
    *out = e;
    return DIR_OK;
    DIR_CATCH
}

int dir_engine_destroy(dir_engine* e) {
    DIR_TRY
    if (!e) return DIR_OK;
    e->release();
    for (ProfSlot& s : e->prof) {
        (void)hipEventDestroy(s.start);
        (void)hipEventDestroy(s.stop);
    }
    delete e;
    return DIR_OK;
    DIR_CATCH
}

int dir_engine_set_tensor(dir_engine* e, const char* key, const float* data, const int64_t* shape,
                          int ndim) {
    DIR_TRY
    if (!e || !key || !data || ndim < 0 || (ndim > 0 && !shape))
        return fail(DIR_ERR_INVALID, "set_tensor: null argument");
    std::string k(key);
    if (k.rfind("module.", 0) == 0) k = k.substr(7);  // DataParallel prefix, common.py:128-131
    if (k.size() >= 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0) return DIR_OK;
    HostTensor t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0) return fail(DIR_ERR_INVALID, "set_tensor: negative dimension");
        t.shape.push_back(shape[i]);
        n *= shape[i];
    }
    t.data.assign(data, data + n);
    e->state[k] = std::move(t);
    e->finalized = false;
    return DIR_OK;
    DIR_CATCH
}

int dir_engine_finalize(dir_engine* e, int dtype) {
    DIR_TRY
    if (!e) return fail(DIR_ERR_INVALID, "finalize: null engine");
    return e->finalize(dtype);
    DIR_CATCH
}

int dir_engine_out_dim(const dir_engine* e, int* out_dim) {
    if (!e || !out_dim) return fail(DIR_ERR_INVALID, "out_dim: null argument");
    *out_dim = e->desc.without_fc ? e->head_dim : e->desc.out_dim;
    return DIR_OK;
}

int dir_workspace_bytes(const dir_engine* e, int B, int H, int W, size_t* bytes) {
    DIR_TRY
    if (!e || !bytes) return fail(DIR_ERR_INVALID, "workspace_bytes: null argument");
    Plan p;
    int rc = e->plan(B, H, W, &p);
    if (rc != DIR_OK) return rc;
    *bytes = p.total;
    return DIR_OK;
    DIR_CATCH
}

int dir_forward(dir_engine* e, const void* img, int B, int H, int W, int fmt, float* desc_out,
                void* ws, size_t ws_bytes, void* stream) {
    DIR_TRY
    if (!e || !img || !desc_out) return fail(DIR_ERR_INVALID, "forward: null argument");
    return e->forward(img, B, H, W, fmt, desc_out, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes,
                      (hipStream_t)stream);
    DIR_CATCH
}

int dir_conv_bn_act_f32(const float* x, const float* w, const float* bias, const float* res, float* y, int B, int H,
                        int W, int Cin, int Cout, int R, int S, int stride, int pad, int OH, int OW, int relu,
                        void* stream) {
    DIR_TRY
    if (!x || !w || !bias || !y) return fail(DIR_ERR_INVALID, "conv_bn_act_f32: null argument");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0)
        return fail(DIR_ERR_INVALID, "conv_bn_act_f32: bad dimension");     // (before the geometry check divides by stride)
    if (OH != (H + 2 * pad - R) / stride + 1 || OW != (W + 2 * pad - S) / stride + 1)
        return fail(DIR_ERR_INVALID, "conv_bn_act_f32: OH/OW do not match the conv geometry");
    ConvF32Args a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w; a.bias = bias; a.res = res; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout;
    a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.relu = relu;
    a.M = B * OH * OW;
    a.Ktot = R * S * Cin;
    return conv_f32_launch(a, (hipStream_t)stream);
    DIR_CATCH
}

int dir_conv_bn_act_pair(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                         const void* res_hi, const void* res_lo, void* y_hi, void* y_lo, int B, int H, int W, int Cin,
                         int Cout, int R, int S, int stride, int pad, int OH, int OW, int relu, void* stream) {
    DIR_TRY
    if (!x_hi || !w_hi || !w_lo || !bias || !y_hi) return fail(DIR_ERR_INVALID, "conv_bn_act_pair: null argument");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0)
        return fail(DIR_ERR_INVALID, "conv_bn_act_pair: bad dimension");
    if (OH != (H + 2 * pad - R) / stride + 1 || OW != (W + 2 * pad - S) / stride + 1 || OH <= 0 || OW <= 0)
        return fail(DIR_ERR_INVALID, "conv_bn_act_pair: OH/OW do not match the conv geometry");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const uint16_t*)x_hi; a.x_lo = (const uint16_t*)x_lo;
    a.w = (const uint16_t*)w_hi; a.w_lo = (const uint16_t*)w_lo;
    a.bias = bias;
    a.res = (const uint16_t*)res_hi; a.res_lo = (const uint16_t*)res_lo;
    a.y = (uint16_t*)y_hi; a.y_lo = (uint16_t*)y_lo;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout;
    a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.relu = relu ? 1 : 0;
    a.M = B * OH * OW;
    a.Ktot = R * S * Cin;
    return conv_pair_launch(a, (hipStream_t)stream);
    DIR_CATCH
}

int dir_conv_pair_dual(const void* t2_hi, const void* t2_lo, const void* x_hi, const void* x_lo, const void* wcat_hi,
                       const void* wcat_lo, const float* bias, void* y_hi, void* y_lo, int B, int H, int W, int Cin,
                       int Cout, int relu, void* stream) {
    DIR_TRY
    if (!t2_hi || !t2_lo || !x_hi || !x_lo || !wcat_hi || !wcat_lo || !bias || !y_hi)
        return fail(DIR_ERR_INVALID, "conv_pair_dual: null argument");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return fail(DIR_ERR_INVALID, "conv_pair_dual: bad dimension");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const uint16_t*)t2_hi; a.x_lo = (const uint16_t*)t2_lo;
    a.x2 = (const uint16_t*)x_hi; a.x2_lo = (const uint16_t*)x_lo;
    a.w = (const uint16_t*)wcat_hi; a.w_lo = (const uint16_t*)wcat_lo;
    a.bias = bias;
    a.y = (uint16_t*)y_hi; a.y_lo = (uint16_t*)y_lo;
    a.B = B; a.H = a.OH = H; a.W = a.OW = W; a.Cin = a.Cin2 = Cin; a.Cout = Cout;
    a.R = a.S = 1; a.stride = 1; a.relu = relu ? 1 : 0;
    a.M = B * H * W;
    a.Ktot = 2 * Cin;
    return conv_pair_launch(a, (hipStream_t)stream);
    DIR_CATCH
}

int dir_prep_input_pair(const void* img, int img_format, const float* mean3, const float* std3, void* out_hi,
                        void* out_lo, int B, int H, int W, void* stream) {
    DIR_TRY
    if (B <= 0 || H <= 0 || W <= 0) return fail(DIR_ERR_INVALID, "prep_input_pair: bad dimension");
    return prep_input_pair(img, img_format, mean3, std3, out_hi, out_lo, B, H, W, (hipStream_t)stream);
    DIR_CATCH
}

int dir_stem_pool_pair(const void* s2d_hi, const void* s2d_lo, const void* w_hi, const void* w_lo, const float* bias,
                       void* y_hi, void* y_lo, int B, int H2, int W2, int OH, int OW, void* stream) {
    DIR_TRY
    if (!s2d_hi || !s2d_lo || !w_hi || !w_lo || !bias || !y_hi || !y_lo) return fail(DIR_ERR_INVALID, "stem_pool_pair: null pointer");
    if (B <= 0 || H2 <= 0 || W2 <= 0 || OH <= 0 || OW <= 0) return fail(DIR_ERR_INVALID, "stem_pool_pair: bad dimension");
    // the 7x7 s2 p3 conv of an H x W image has (H - 1) / 2 + 1 = (H + 1) / 2 output rows - the rows of its space-to-depth
    // grid, for even and odd H alike; any other OH / OW would size the grid and the stores beyond the caller's buffers
    if (OH != H2 || OW != W2)
        return fail(DIR_ERR_INVALID, "stem_pool_pair: OH x OW must equal the space-to-depth grid H2 x W2 (conv 7x7 s2 p3)");
    return stem_pool_pair_launch(s2d_hi, s2d_lo, w_hi, w_lo, bias, y_hi, y_lo, B, H2, W2, OH, OW, (hipStream_t)stream);
    DIR_CATCH
}

int dir_stem_pool_u8(const void* img_u8, const float* w_oihw, const float* bn_scale, const float* bn_bias, const float* mean3,
                     const float* std3, void* s2d_ws, void* y_hi, void* y_lo, int B, int H, int W, int seg_tiles, void* stream) {
    DIR_TRY
    if (!img_u8 || !w_oihw || !bn_scale || !bn_bias || !mean3 || !std3 || !s2d_ws || !y_hi || !y_lo)
        return fail(DIR_ERR_INVALID, "stem_pool_u8: null pointer");
    if (B <= 0 || H < 7 || W < 7) return fail(DIR_ERR_INVALID, "stem_pool_u8: bad dimension (the image must hold the 7x7 filter)");
    std::vector<uint16_t> hi, lo;
    std::vector<float> b2, corr;
    int rc = fold_stem_u8(w_oihw, bn_scale, bn_bias, mean3, std3, hi, lo, b2, corr);
    if (rc != DIR_OK) return rc;
    // (a per-call upload: this is the parity entry point; the engine folds once at finalize and keeps the tables)
    char* d = nullptr;
    const size_t nw = hi.size() * 2, nb = b2.size() * 4, nc = corr.size() * 4;
    DIR_HIP_CHECK(hipMalloc((void**)&d, 2 * nw + nb + nc));
    hipError_t e = hipMemcpy(d, hi.data(), nw, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + nw, lo.data(), nw, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 2 * nw, b2.data(), nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 2 * nw + nb, corr.data(), nc, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail(DIR_ERR_HIP, std::string("stem_pool_u8 upload: ") + hipGetErrorString(e));
    }
    const bool raw = stem_pool_u8_raw_ok(img_u8, B, H, W);
    rc = raw ? DIR_OK : prep_input_u8(img_u8, s2d_ws, B, H, W, (hipStream_t)stream);
    if (rc == DIR_OK)
        rc = stem_pool_u8_launch(raw ? img_u8 : nullptr, s2d_ws, d, d + nw, (const float*)(d + 2 * nw), (const float*)(d + 2 * nw + nb), y_hi, y_lo, B, H, W,
                                 (hipStream_t)stream, nullptr, seg_tiles);
    e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d);
    if (rc != DIR_OK) return rc;
    if (e != hipSuccess) return fail(DIR_ERR_HIP, std::string("stem_pool_u8: ") + hipGetErrorString(e));
    return DIR_OK;
    DIR_CATCH
}

int dir_engine_overflow(dir_engine* e, void* stream, int* overflowed) {
    DIR_TRY
    if (!e || !overflowed) return fail(DIR_ERR_INVALID, "overflow: null argument");
    return e->overflow((hipStream_t)stream, overflowed);
    DIR_CATCH
}

int dir_forward_features(dir_engine* e, const void* img, int B, int H, int W, int fmt,
                         void* feat_out, int* h, int* w, int* c, void* ws, size_t ws_bytes,
                         void* stream) {
    DIR_TRY
    if (!e || !img || !feat_out) return fail(DIR_ERR_INVALID, "forward_features: null argument");
    return e->forward(img, B, H, W, fmt, nullptr, feat_out, h, w, c, ws, ws_bytes,
                      (hipStream_t)stream);
    DIR_CATCH
}

int dir_engine_autotune(dir_engine* e, int B, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    DIR_TRY
    if (!e) return fail(DIR_ERR_INVALID, "autotune: null engine");
    if (e->dtype == DIR_F32) return DIR_OK;   // the strict path has one kernel per conv: nothing to choose
    const bool was_prof = e->profiling;
    e->profiling = false;
    e->tuning = true;
    int rc = e->forward(nullptr, B, H, W, DIR_IMG_F32_NCHW, nullptr, nullptr, nullptr, nullptr,
                        nullptr, ws, ws_bytes, (hipStream_t)stream);
    e->tuning = false;
    e->profiling = was_prof;
    if (rc != DIR_OK) return rc;
    DIR_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return DIR_OK;
    DIR_CATCH
}

int dir_engine_tuning_export(const dir_engine* e, char* buf, size_t cap, size_t* needed) {
    DIR_TRY
    if (!e || !needed) return fail(DIR_ERR_INVALID, "tuning_export: null argument");
    std::string out;
    for (const ConvLayer& L : e->convs)
        for (const auto& kv : L.tuned)
            out += L.name + " " + std::to_string(kv.first) + " " + conv_variant(kv.second).name + "\n";
    *needed = out.size() + 1;
    if (buf && cap >= out.size() + 1) memcpy(buf, out.c_str(), out.size() + 1);
    return DIR_OK;
    DIR_CATCH
}

int dir_engine_tuning_import(dir_engine* e, const char* text) {
    DIR_TRY
    if (!e || !text) return fail(DIR_ERR_INVALID, "tuning_import: null argument");
    std::string s(text);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t eol = s.find('\n', pos);
        if (eol == std::string::npos) eol = s.size();
        const std::string line = s.substr(pos, eol - pos);
        pos = eol + 1;
        if (line.empty()) continue;
        const size_t a = line.find(' '), b = line.rfind(' ');
        if (a == std::string::npos || b == a) return fail(DIR_ERR_INVALID, "tuning_import: bad line: " + line);
        const std::string lname = line.substr(0, a), vname = line.substr(b + 1);
        const long M = atol(line.substr(a + 1, b - a - 1).c_str());
        int variant = -1;
        for (int v = 0; v < conv_variant_count(); ++v)
            if (vname == conv_variant(v).name) variant = v;
        if (variant < 0) continue;   // a variant this build no longer has: fall back to the heuristic
        for (ConvLayer& L : e->convs)
            if (L.name == lname) L.tuned[M] = variant;
    }
    return DIR_OK;
    DIR_CATCH
}

int dir_engine_set_profiling(dir_engine* e, int enabled) {
    if (!e) return fail(DIR_ERR_INVALID, "set_profiling: null engine");
    e->profiling = enabled != 0;
    e->prof_paused = false;
    e->prof_used = 0;
    // enabled > 1: pre-create that many event pairs so none is created inside a timed region
    while ((int)e->prof.size() < enabled && enabled > 1) {
        ProfSlot s;
        DIR_HIP_CHECK(hipEventCreate(&s.start));
        DIR_HIP_CHECK(hipEventCreate(&s.stop));
        e->prof.push_back(s);
    }
    return DIR_OK;
}

int dir_engine_profile_pause(dir_engine* e, int paused) {
    if (!e) return fail(DIR_ERR_INVALID, "profile_pause: null engine");
    e->prof_paused = paused != 0;
    return DIR_OK;
}

int dir_engine_get_profile(dir_engine* e, dir_prof_record* out, int cap, int* n) {
    DIR_TRY
    if (!e || !n) return fail(DIR_ERR_INVALID, "get_profile: null argument");
    const int have = (int)e->prof_used;
    for (int i = 0; i < have; ++i) {
        ProfSlot& s = e->prof[i];
        DIR_HIP_CHECK(hipEventSynchronize(s.stop));
        if (out && i < cap) {
            float ms = 0.f;
            DIR_HIP_CHECK(hipEventElapsedTime(&ms, s.start, s.stop));
            dir_prof_record& r = out[i];
            memset(&r, 0, sizeof(r));
            strncpy(r.name, s.name.c_str(), sizeof(r.name) - 1);
            strncpy(r.kernel, s.kernel.c_str(), sizeof(r.kernel) - 1);
            r.flops = s.flops;
            r.bytes = s.bytes;
            r.ms = ms;
        }
    }
    *n = have;
    e->prof_used = 0;
    return DIR_OK;
    DIR_CATCH
}

// ---- per-op entry points --------------------------------------------------------------------------
int dir_conv_variant_count(void) { return conv_variant_count(); }
int dir_conv_variant_name(int variant, char* buf, int cap) {
    if (variant < 0 || variant >= conv_variant_count() || !buf || cap <= 0)
        return fail(DIR_ERR_INVALID, "variant_name: bad argument");
    strncpy(buf, conv_variant(variant).name, cap - 1);
    buf[cap - 1] = 0;
    return DIR_OK;
}

static int fill_conv_args(ConvArgs& a, const void* x, const void* w, const float* bias,
                          const void* res, void* y, int B, int H, int W, int Cin, int Cout, int R,
                          int S, int stride, int pad, int OH, int OW, int relu);

int dir_conv_heuristic(int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int OH,
                       int OW, int has_residual, char* name, int cap, int* ksplit) {
    DIR_TRY
    if (!name || cap <= 0) return fail(DIR_ERR_INVALID, "conv_heuristic: null name buffer");
    ConvArgs a;
    // host-only decision: the pointers are never dereferenced, they only have to be non-null
    static const float dummy = 0.f;
    int rc = fill_conv_args(a, &dummy, &dummy, &dummy, has_residual ? &dummy : nullptr, (void*)&dummy, B, H, W, Cin,
                            Cout, R, S, stride, pad, OH, OW, 1);
    if (rc != DIR_OK) return rc;
    const int v = conv_pick_variant(a);
    if (v < 0) return fail(DIR_ERR_INVALID, "conv_heuristic: no admissible variant for this shape");
    strncpy(name, conv_variant(v).name, cap - 1);
    name[cap - 1] = 0;
    if (ksplit) *ksplit = conv_splitk_factor(v, a);
    return DIR_OK;
    DIR_CATCH
}

int dir_conv_variant_admissible(int variant, int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                int OH, int OW, int has_residual, int* admissible) {
    DIR_TRY
    if (!admissible) return fail(DIR_ERR_INVALID, "conv_variant_admissible: null result pointer");
    if (variant < 0 || variant >= conv_variant_count()) return fail(DIR_ERR_INVALID, "conv_variant_admissible: no such variant");
    ConvArgs a;
    static const float dummy = 0.f;   // host-only decision: the pointers are never dereferenced
    int rc = fill_conv_args(a, &dummy, &dummy, &dummy, has_residual ? &dummy : nullptr, (void*)&dummy, B, H, W, Cin,
                            Cout, R, S, stride, pad, OH, OW, 1);
    if (rc != DIR_OK) return rc;
    *admissible = conv_variant_admissible(variant, a) ? 1 : 0;
    return DIR_OK;
    DIR_CATCH
}

static int fill_conv_args(ConvArgs& a, const void* x, const void* w, const float* bias,
                          const void* res, void* y, int B, int H, int W, int Cin, int Cout, int R,
                          int S, int stride, int pad, int OH, int OW, int relu) {
    if (!x || !w || !bias || !y) return fail(DIR_ERR_INVALID, "conv: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 ||
        pad < 0 || OH <= 0 || OW <= 0)
        return fail(DIR_ERR_INVALID, "conv: bad dimension");
    memset(&a, 0, sizeof(a));
    a.x = (const uint16_t*)x;
    a.w = (const uint16_t*)w;
    a.bias = bias;
    a.res = (const uint16_t*)res;
    a.y = (uint16_t*)y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout;
    a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.relu = relu ? 1 : 0;
    a.M = B * OH * OW;
    a.Ktot = R * S * Cin;
    a.T = a.Ktot / 64;
    return DIR_OK;
}

int dir_conv_bn_act(const void* x, const void* w, const float* bias, const void* res, void* y,
                    int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                    int OH, int OW, int relu, int dtype, int variant, void* stream) {
    DIR_TRY
    ConvArgs a;
    int rc = fill_conv_args(a, x, w, bias, res, y, B, H, W, Cin, Cout, R, S, stride, pad, OH, OW, relu);
    if (rc != DIR_OK) return rc;
    return conv_launch(a, dtype, variant, (hipStream_t)stream);
    DIR_CATCH
}

int dir_conv_bn_act_splitk(const void* x, const void* w, const float* bias, const void* res, void* y,
                           int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                           int OH, int OW, int relu, int dtype, int variant, int ksplit, void* scratch,
                           size_t scratch_bytes, int* ksplit_used, void* stream) {
    DIR_TRY
    ConvArgs a;
    int rc = fill_conv_args(a, x, w, bias, res, y, B, H, W, Cin, Cout, R, S, stride, pad, OH, OW, relu);
    if (rc != DIR_OK) return rc;
    if (variant < 0) variant = conv_pick_variant(a);
    a.ksplit = ksplit < 0 ? conv_splitk_factor(variant, a) : ksplit;
    a.partial = (float*)scratch;
    if (a.ksplit > 1 && conv_splitk_bytes(a, a.ksplit) > scratch_bytes)
        return fail(DIR_ERR_WORKSPACE, "conv: split-K scratch too small");
    if (ksplit_used) *ksplit_used = a.ksplit > 1 ? a.ksplit : 1;
    return conv_launch(a, dtype, variant, (hipStream_t)stream);
    DIR_CATCH
}

int dir_conv_c3c1(const void* t2, const void* w3, const float* bias3, const void* res, void* y, const void* w1,
                  const float* bias1, void* t1, int B, int H, int W, int P, int P2, int relu3, int relu1, int dtype,
                  void* stream) {
    DIR_TRY
    ConvArgs a;
    int rc = fill_conv_args(a, t2, w3, bias3, res, y, B, H, W, P, 4 * P, 1, 1, 1, 0, H, W, relu3);
    if (rc != DIR_OK) return rc;
    if (!w1 || !bias1 || !t1 || !res) return fail(DIR_ERR_INVALID, "conv_c3c1: null pointer");
    if (dtype != DIR_BF16 && dtype != DIR_FP16) return fail(DIR_ERR_INVALID, "conv_c3c1: bad dtype");
    a.w2 = (const uint16_t*)w1;
    a.bias2 = bias1;
    a.y2 = (uint16_t*)t1;
    a.Cout2 = P2;
    a.relu2 = relu1 ? 1 : 0;
    if (!conv_c3c1_admissible(a))
        return fail(DIR_ERR_INVALID, "conv_c3c1: planes must be 64, 128 (P2 = P, or 128 after 64) or 256 (P2 = 256, B*H*W % 64 == 0)");
    if (((uintptr_t)t2 & 15) || ((uintptr_t)w3 & 15) || ((uintptr_t)res & 15) || ((uintptr_t)y & 15) ||
        ((uintptr_t)w1 & 15) || ((uintptr_t)t1 & 15) || ((uintptr_t)bias3 & 15) || ((uintptr_t)bias1 & 15))
        return fail(DIR_ERR_INVALID, "conv_c3c1: tensors must be 16-byte aligned");
    hipError_t e = conv_c3c1_launch(a, dtype, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DIR_ERR_HIP, std::string("conv_c3c1 launch: ") + hipGetErrorString(e));
    return DIR_OK;
    DIR_CATCH
}

int dir_conv_c3c1_ds(const void* t2, const void* x, const void* wcat, const float* bias, void* y, const void* w1,
                     const float* bias1, void* t1, int B, int H, int W, int relu3, int relu1, int dtype,
                     void* stream) {
    DIR_TRY
    ConvArgs a;
    int rc = fill_conv_args(a, t2, wcat, bias, nullptr, y, B, H, W, 64, 256, 1, 1, 1, 0, H, W, relu3);
    if (rc != DIR_OK) return rc;
    if (!x || !w1 || !bias1 || !t1) return fail(DIR_ERR_INVALID, "conv_c3c1_ds: null pointer");
    if (dtype != DIR_BF16 && dtype != DIR_FP16) return fail(DIR_ERR_INVALID, "conv_c3c1_ds: bad dtype");
    a.x2 = (const uint16_t*)x;
    a.Cin2 = 64;
    a.w2 = (const uint16_t*)w1;
    a.bias2 = bias1;
    a.y2 = (uint16_t*)t1;
    a.Cout2 = 64;
    a.relu2 = relu1 ? 1 : 0;
    if (!conv_c3c1_admissible(a)) return fail(DIR_ERR_INVALID, "conv_c3c1_ds: shape not admissible");
    if (((uintptr_t)t2 & 15) || ((uintptr_t)x & 15) || ((uintptr_t)wcat & 15) || ((uintptr_t)y & 15) ||
        ((uintptr_t)w1 & 15) || ((uintptr_t)t1 & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)bias1 & 15))
        return fail(DIR_ERR_INVALID, "conv_c3c1_ds: tensors must be 16-byte aligned");
    hipError_t e = conv_c3c1_launch(a, dtype, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DIR_ERR_HIP, std::string("conv_c3c1_ds launch: ") + hipGetErrorString(e));
    return DIR_OK;
    DIR_CATCH
}

int dir_conv_c3c1_wpair(const void* t2, const void* w3, const void* w3_lo, const float* bias3, const void* res, void* y,
                        const void* w1, const void* w1_lo, const float* bias1, void* t1, int B, int H, int W, int P2,
                        int relu3, int relu1, void* stream) {
    DIR_TRY
    ConvArgs a;
    int rc = fill_conv_args(a, t2, w3, bias3, res, y, B, H, W, 64, 256, 1, 1, 1, 0, H, W, relu3);
    if (rc != DIR_OK) return rc;
    if (!w3_lo || !w1 || !bias1 || !t1 || !res) return fail(DIR_ERR_INVALID, "conv_c3c1_wpair: null pointer");
    a.w_lo = (const uint16_t*)w3_lo;
    a.w2 = (const uint16_t*)w1;
    a.w2_lo = (const uint16_t*)w1_lo;
    a.bias2 = bias1;
    a.y2 = (uint16_t*)t1;
    a.Cout2 = P2;
    a.relu2 = relu1 ? 1 : 0;
    if (!conv_c3c1_admissible(a))
        return fail(DIR_ERR_INVALID, "conv_c3c1_wpair: planes 64; P2 = 64 (w1_lo required) or 128 (w1_lo optional)");
    const void* ptrs[] = {t2, w3, w3_lo, res, y, w1, w1_lo, t1, bias3, bias1};
    for (const void* q : ptrs)
        if ((uintptr_t)q & 15) return fail(DIR_ERR_INVALID, "conv_c3c1_wpair: tensors must be 16-byte aligned");
    hipError_t e = conv_c3c1_launch(a, DIR_FP16, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DIR_ERR_HIP, std::string("conv_c3c1_wpair launch: ") + hipGetErrorString(e));
    return DIR_OK;
    DIR_CATCH
}

int dir_conv_c3c1_ds_wpair(const void* t2, const void* x, const void* x_lo, const void* wcat, const void* wcat_lo,
                           const float* bias, void* y, const void* w1, const void* w1_lo, const float* bias1, void* t1,
                           int B, int H, int W, int relu3, int relu1, void* stream) {
    DIR_TRY
    ConvArgs a;
    int rc = fill_conv_args(a, t2, wcat, bias, nullptr, y, B, H, W, 64, 256, 1, 1, 1, 0, H, W, relu3);
    if (rc != DIR_OK) return rc;
    if (!x || !x_lo || !wcat_lo || !w1 || !w1_lo || !bias1 || !t1) return fail(DIR_ERR_INVALID, "conv_c3c1_ds_wpair: null pointer");
    a.x2 = (const uint16_t*)x;
    a.x2_lo = (const uint16_t*)x_lo;
    a.Cin2 = 64;
    a.w_lo = (const uint16_t*)wcat_lo;
    a.w2 = (const uint16_t*)w1;
    a.w2_lo = (const uint16_t*)w1_lo;
    a.bias2 = bias1;
    a.y2 = (uint16_t*)t1;
    a.Cout2 = 64;
    a.relu2 = relu1 ? 1 : 0;
    if (!conv_c3c1_admissible(a)) return fail(DIR_ERR_INVALID, "conv_c3c1_ds_wpair: shape not admissible");
    const void* ptrs[] = {t2, x, x_lo, wcat, wcat_lo, y, w1, w1_lo, t1, bias, bias1};
    for (const void* q : ptrs)
        if ((uintptr_t)q & 15) return fail(DIR_ERR_INVALID, "conv_c3c1_ds_wpair: tensors must be 16-byte aligned");
    hipError_t e = conv_c3c1_launch(a, DIR_FP16, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DIR_ERR_HIP, std::string("conv_c3c1_ds_wpair launch: ") + hipGetErrorString(e));
    return DIR_OK;
    DIR_CATCH
}

int dir_conv_dual(const void* t2, const void* x, const void* wcat, const float* bias, void* y, int B, int OH, int OW,
                  int Cin, int Cout, int Cin2, int H2, int W2, int stride2, int relu, int dtype, void* stream) {
    DIR_TRY
    ConvArgs a;
    int rc = fill_conv_args(a, t2, wcat, bias, nullptr, y, B, OH, OW, Cin, Cout, 1, 1, 1, 0, OH, OW, relu);
    if (rc != DIR_OK) return rc;
    if (!x || Cin2 <= 0 || H2 <= 0 || W2 <= 0 || stride2 <= 0) return fail(DIR_ERR_INVALID, "conv_dual: bad argument");
    if ((OH - 1) * stride2 >= H2 || (OW - 1) * stride2 >= W2)
        return fail(DIR_ERR_INVALID, "conv_dual: the strided pixel map leaves the second source");
    if (Cin % 64 || Cin2 % 64) return fail(DIR_ERR_INVALID, "conv_dual: channel counts must be multiples of 64");
    a.x2 = (const uint16_t*)x;
    a.Cin2 = Cin2;
    a.H2 = H2;
    a.W2 = W2;
    a.stride2 = stride2;
    a.Ktot = Cin + Cin2;
    a.T = a.Ktot / 64;
    if (((uintptr_t)t2 & 15) || ((uintptr_t)wcat & 15) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15))
        return fail(DIR_ERR_INVALID, "conv_dual: tensors must be 16-byte aligned");
    if ((long)a.M * Cout >= (1L << 30) || (long)a.M * Cin >= (1L << 30))
        return fail(DIR_ERR_INVALID, "conv_dual: tensor exceeds 2^31 bytes; lower the batch");
    if (dtype != DIR_BF16 && dtype != DIR_FP16) return fail(DIR_ERR_INVALID, "conv_dual: bad dtype");
    // (small shapes included: the op-level entry point runs the form wherever the tile divides Cout)
    // what the engine runs at batch size: the persistent two-source ring (conv_persist.hip DUAL)
    const int variant = conv_pick_dual_variant(a, true);
    if (variant < 0) return fail(DIR_ERR_INVALID, "conv_dual: Cout must be a multiple of 256");
    return conv_launch(a, dtype, variant, (hipStream_t)stream);
    DIR_CATCH
}

int dir_conv_bn_act_naive(const void* x, const void* w, const float* bias, const void* res, void* y,
                          int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                          int OH, int OW, int relu, int dtype, void* stream) {
    DIR_TRY
    ConvArgs a;
    int rc = fill_conv_args(a, x, w, bias, res, y, B, H, W, Cin, Cout, R, S, stride, pad, OH, OW, relu);
    if (rc != DIR_OK) return rc;
    if (dtype != DIR_BF16 && dtype != DIR_FP16) return fail(DIR_ERR_INVALID, "conv: bad dtype");
    return conv_launch_naive(a, dtype, (hipStream_t)stream);
    DIR_CATCH
}

int dir_prep_input(const void* img, int fmt, const float* mean3, const float* std3, void* out,
                   int B, int H, int W, int dtype, void* stream) {
    DIR_TRY
    if (!img || !out || B <= 0 || H <= 0 || W <= 0) return fail(DIR_ERR_INVALID, "prep_input: bad argument");
    return prep_input(img, fmt, mean3, std3, out, B, H, W, dtype, (hipStream_t)stream);
    DIR_CATCH
}

int dir_stem_pool(const void* s2d, const void* w, const float* bias, void* y, int B, int H2, int W2,
                  int OH, int OW, int dtype, void* stream) {
    DIR_TRY
    if (!s2d || !w || !bias || !y || B <= 0 || H2 <= 0 || W2 <= 0 || OH <= 0 || OW <= 0)
        return fail(DIR_ERR_INVALID, "stem_pool: bad argument");
    if (OH != H2 || OW != W2)   // (conv 7x7 s2 p3: (H - 1) / 2 + 1 == (H + 1) / 2 for every H; see dir_stem_pool_pair)
        return fail(DIR_ERR_INVALID, "stem_pool: OH x OW must equal the space-to-depth grid H2 x W2 (conv 7x7 s2 p3)");
    return stem_pool_launch(s2d, w, bias, y, B, H2, W2, OH, OW, dtype, (hipStream_t)stream);
    DIR_CATCH
}

int dir_maxpool_3x3s2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
    DIR_TRY
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(DIR_ERR_INVALID, "maxpool: bad argument");
    return maxpool_3x3s2(x, y, B, H, W, C, dtype, (hipStream_t)stream);
    DIR_CATCH
}

int dir_global_pool(const void* x, float* out, int B, int H, int W, int C, int pooling, float p,
                    float eps, float center_bias, int dtype, void* stream) {
    DIR_TRY
    if (!x || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(DIR_ERR_INVALID, "global_pool: bad argument");
    return global_pool(x, out, C, B, H, W, C, pooling, p, eps, center_bias, dtype, (hipStream_t)stream);
    DIR_CATCH
}

int dir_resize_workspace_bytes(int B, int H, int W, int OH, int OW, size_t* bytes) {
    DIR_TRY
    if (!bytes || B <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0)
        return fail(DIR_ERR_INVALID, "resize_workspace_bytes: bad argument");
    *bytes = resize_workspace_bytes(B, H, W, OH, OW);
    return DIR_OK;
    DIR_CATCH
}

int dir_resize_bilinear_u8(const void* src, void* dst, int B, int H, int W, int OH, int OW, void* workspace,
                           size_t workspace_bytes, void* stream) {
    DIR_TRY
    if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0)
        return fail(DIR_ERR_INVALID, "resize_bilinear_u8: bad argument");
    return resize_bilinear_u8((const uint8_t*)src, (uint8_t*)dst, B, H, W, OH, OW, workspace, workspace_bytes,
                              (hipStream_t)stream);
    DIR_CATCH
}

int dir_upsample_add(const void* x, const void* low, void* y, int B, int H, int W, int h, int w, int C,
                     int dtype, void* stream) {
    DIR_TRY
    if (!x || !low || !y || B <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0 || C <= 0)
        return fail(DIR_ERR_INVALID, "upsample_add: bad argument");
    return upsample_add(x, low, y, B, H, W, h, w, C, dtype, (hipStream_t)stream);
    DIR_CATCH
}

int dir_l2norm_rows(float* x, int rows, int cols, float eps, void* stream) {
    DIR_TRY
    if (!x || rows < 0 || cols <= 0) return fail(DIR_ERR_INVALID, "l2norm_rows: bad argument");
    return l2norm_rows(x, rows, cols, eps, (hipStream_t)stream);
    DIR_CATCH
}

/* host-only: the number of K slices dir_gemm_nt_f32 runs a shape with (1 = no split) */
int dir_gemm_splitk_factor(int NP, int NQ, int K) { return gemm_splitk_factor(NP, NQ, K); }

int dir_gemm_nt_f32(const float* P, int ldp, const float* Q, int ldq, float* out, int ldo, int NP,
                    int NQ, int K, const float* qsub, const float* bias, const float* alpha,
                    void* stream) {
    DIR_TRY
    if (NP < 0 || NQ < 0) return fail(DIR_ERR_INVALID, "gemm_nt_f32: negative size");
    if (NP == 0 || NQ == 0) return DIR_OK;
    if (!P || !Q || !out) return fail(DIR_ERR_INVALID, "gemm_nt_f32: null pointer");
    if (ldo < NP) return fail(DIR_ERR_INVALID, "gemm_nt_f32: ldo < NP");
    return gemm_nt_f32(P, ldp, Q, ldq, out, ldo, NP, NQ, K, qsub, bias, alpha, (hipStream_t)stream);
    DIR_CATCH
}

int dir_fc_l2(const float* x, int B, int K, const float* W, const float* b, int D, float* out,
              void* stream) {
    DIR_TRY
    if (!x || !W || !out || B <= 0 || K <= 0 || D <= 0) return fail(DIR_ERR_INVALID, "fc_l2: bad argument");
    int rc = gemm_nt_f32(W, K, x, K, out, D, D, B, K, nullptr, b, nullptr, (hipStream_t)stream);
    if (rc != DIR_OK) return rc;
    return l2norm_rows(out, B, D, 1e-12f, (hipStream_t)stream);
    DIR_CATCH
}

static int pca_whiten_impl(const float* X, int N, int D, const float* mean, const float* components, int v, const float* scale,
                           int l2norm, float* out, void* stream, bool unit_range) {
    if (N < 0 || D <= 0 || v <= 0) return fail(DIR_ERR_INVALID, "pca_whiten_l2: bad size");
    if (N == 0) return DIR_OK;
    if (!X || !components || !out) return fail(DIR_ERR_INVALID, "pca_whiten_l2: null pointer");
    int rc = DIR_OK;
    bool done = false;
    // Large sets whose operands the caller knows to be bounded (dir_pca_whiten_l2_unit): two fp16 planes per operand, three
    // plane products on the fp16 matrix cores (sim_split.hip whiten_split) - 8.4 TFLOP of fp32 MFMA chain at config D otherwise
    if (unit_range && !dir::env().sim_exact && N >= kSimSplitMinRows && whiten_split_admissible(X, D, N, D, v)) {
        const size_t bytes = whiten_split_workspace_bytes(v, D);
        void* ws = nullptr;
        if (hipMallocAsync(&ws, bytes, (hipStream_t)stream) == hipSuccess) {
            rc = whiten_split(X, D, N, components, D, v, D, mean, scale, out, v, ws, bytes, (hipStream_t)stream);
            const hipError_t fe = hipFreeAsync(ws, (hipStream_t)stream);
            if (rc != DIR_OK) return rc;
            DIR_HIP_CHECK(fe);
            done = true;
        } else {
            (void)hipGetLastError();   // no memory pool / out of memory: the exact chain below needs no scratch
        }
    }
    if (!done) rc = gemm_nt_f32(components, D, X, D, out, v, v, N, D, mean, nullptr, scale, (hipStream_t)stream);
    if (rc != DIR_OK || !l2norm) return rc;
    return l2norm_rows(out, N, v, 1e-12f, (hipStream_t)stream);
}

int dir_pca_whiten_l2(const float* X, int N, int D, const float* mean, const float* components, int v,
                      const float* scale, int l2norm, float* out, void* stream) {
    DIR_TRY
    return pca_whiten_impl(X, N, D, mean, components, v, scale, l2norm, out, stream, false);
    DIR_CATCH
}

int dir_pca_whiten_l2_unit(const float* X, int N, int D, const float* mean, const float* components, int v,
                           const float* scale, int l2norm, float* out, void* stream) {
    DIR_TRY
    return pca_whiten_impl(X, N, D, mean, components, v, scale, l2norm, out, stream, true);
    DIR_CATCH
}

static int similarity_impl(const float* queries, int Q, const float* database, int N, int D, float* scores, void* stream,
                           bool unit_range) {
    if (Q < 0 || N < 0 || D <= 0) return fail(DIR_ERR_INVALID, "similarity: bad size");
    if (Q == 0 || N == 0) return DIR_OK;
    if (!queries || !database || !scores) return fail(DIR_ERR_INVALID, "similarity: null pointer");
    // Large databases (the 10^6-distractor protocol): three-plane bf16 split on the matrix cores, fp32-accurate
    // products, ~2x the speed of the exact fp32 MFMA chain (sim_split.hip) - or, for operands the caller knows to be
    // bounded (dir_similarity_unit), two fp16 planes and half the products.  Small ones, odd widths and
    // DIRTORCH_AMD_SIM_EXACT=1 keep the k-ordered fmaf chain of gemm_nt_f32.
    const bool exact = dir::env().sim_exact;
    if (!exact && N >= kSimSplitMinRows && similarity_split_admissible(database, D, queries, D, N, Q, D)) {
        const size_t bytes = similarity_split_workspace_bytes(Q, D);
        void* ws = nullptr;
        if (hipMallocAsync(&ws, bytes, (hipStream_t)stream) == hipSuccess) {   // stream-ordered: freed after the kernels
            const int rc = similarity_split(database, D, queries, D, scores, N, N, Q, D, ws, bytes, (hipStream_t)stream,
                                            unit_range);
            const hipError_t fe = hipFreeAsync(ws, (hipStream_t)stream);
            if (rc != DIR_OK) return rc;
            DIR_HIP_CHECK(fe);
            return DIR_OK;
        }
        (void)hipGetLastError();   // no memory pool / out of memory: the exact chain below needs no scratch
    }
    return gemm_nt_f32(database, D, queries, D, scores, N, N, Q, D, nullptr, nullptr, nullptr,
                       (hipStream_t)stream);
}

int dir_similarity(const float* queries, int Q, const float* database, int N, int D, float* scores,
                   void* stream) {
    DIR_TRY
    return similarity_impl(queries, Q, database, N, D, scores, stream, false);
    DIR_CATCH
}

int dir_similarity_unit(const float* queries, int Q, const float* database, int N, int D, float* scores,
                        void* stream) {
    DIR_TRY
    return similarity_impl(queries, Q, database, N, D, scores, stream, true);
    DIR_CATCH
}

int dir_rank_counts(const float* scores, int lds, int Q, int N, const int* probe_idx, int P,
                    int* counts, float* probe_scores, void* stream) {
    DIR_TRY
    if (Q < 0 || N < 0 || P < 0) return fail(DIR_ERR_INVALID, "rank_counts: negative size");
    if (Q == 0 || N == 0 || P == 0) return DIR_OK;
    if (!scores || !probe_idx || !counts || !probe_scores)
        return fail(DIR_ERR_INVALID, "rank_counts: null pointer");
    return rank_counts(scores, lds, Q, N, probe_idx, P, counts, probe_scores, (hipStream_t)stream);
    DIR_CATCH
}

int dir_revisitop_ap(const int* probe_idx, int Q, int P, const int* counts, const float* probe_scores,
                     const int* pos_off, const int* pos_list, const int* junk_off, const int* junk_list,
                     int modes, double* terms, double* ap_out, void* stream) {
    DIR_TRY
    if (Q < 0 || P < 0 || modes < 0) return fail(DIR_ERR_INVALID, "revisitop_ap: negative size");
    if (Q == 0 || modes == 0) return DIR_OK;
    if (!probe_idx || !counts || !probe_scores || !pos_off || !pos_list || !junk_off || !junk_list || !terms ||
        !ap_out)
        return fail(DIR_ERR_INVALID, "revisitop_ap: null pointer");
    return revisitop_ap(probe_idx, Q, P, counts, probe_scores, pos_off, pos_list, junk_off, junk_list, modes,
                        terms, ap_out, (hipStream_t)stream);
    DIR_CATCH
}

int dir_expand_descriptors(const float* descs, int n, const float* db, int m, int D, int k, float alpha,
                           int self_set, float* out, float* sim, size_t sim_bytes, void* stream) {
    DIR_TRY
    if (n < 0 || m < 0 || D <= 0) return fail(DIR_ERR_INVALID, "expand_descriptors: bad size");
    if (n == 0) return DIR_OK;
    if (!descs || !db || !out || !sim) return fail(DIR_ERR_INVALID, "expand_descriptors: null pointer");
    return expand_descriptors(descs, n, db, m, D, k, alpha, self_set, out, sim, sim_bytes, (hipStream_t)stream);
    DIR_CATCH
}

int dir_multiscale_pool(const float* x, float* out, int S, int N, int D, int mode, float gemp,
                        void* stream) {
    DIR_TRY
    if (!x || !out || N < 0 || D < 0) return fail(DIR_ERR_INVALID, "multiscale_pool: bad argument");
    return multiscale_pool(x, out, S, N, D, mode, gemp, (hipStream_t)stream);
    DIR_CATCH
}

}  // extern "C"
