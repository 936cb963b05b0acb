// C ABI (include/w2v2.h): model lifetime, variable I/O, the forward orchestration,
// profiling and the single-operator entry points.  Host code only; the kernels
// live in the sibling .hip files.
//
// Forward order follows the reference exactly: Wav2Vec2ForCTC.call
// (modeling.py:239-255) -> Wav2Vec2Model.call (modeling.py:169-209) ->
// FeatureExtractorLayer x7 (feature_extractor.py:54-59) -> FeatureProjection
// (feature_extractor.py:92-95) -> Wav2Vec2Encoder.call (encoder.py:251-276) ->
// TransformerLayer.call (encoder.py:111-134) -> lm_head.
#include <mutex>
#include <atomic>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "model.h"

namespace w2v2 {

// ---- errors ---------------------------------------------------------------
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* family_name(int f) {
    static const char* names[FAM_COUNT] = {"conv0_stats", "conv0_apply", "gemm_f32", "gemm_bf16", "gemm_split", "layer_norm",
                                           "pos_conv",    "attention",   "ctc",      "misc",      "dropout",    "reduce",
                                           "layer_norm_bwd", "optimizer"};
    return (f >= 0 && f < FAM_COUNT) ? names[f] : "?";
}

// ---- kernel-launch counters (common.h: W2V2_LAUNCH) ---------------------------
thread_local int tl_launch_family = FAM_MISC;
thread_local int tl_prof_depth = 0;
thread_local Profiler* tl_step_prof = nullptr;
static std::atomic<int64_t> g_kernel_launches[FAM_COUNT];
void note_kernel_launch() {
    const int f = tl_launch_family;
    g_kernel_launches[(f >= 0 && f < FAM_COUNT) ? f : FAM_MISC].fetch_add(1, std::memory_order_relaxed);
}
int64_t kernel_launches(int family) {
    return (family >= 0 && family < FAM_COUNT) ? g_kernel_launches[family].load(std::memory_order_relaxed) : 0;
}

// ---- profiler ---------------------------------------------------------------
struct ProfRec {
    int family;
    double flops, bytes;
    hipEvent_t e0, e1;
};
struct Profiler {
    bool enabled = false;
    unsigned mask = 0xFFFFFFFFu;    // families that get an event pair
    int stride = 1;                 // of a family's launches, every stride-th gets the pair (w2v2_profile_sampling)
    int64_t seen[32] = {0};         // launches of each family since the last reset, sampled or not
    int64_t launch_base[32] = {0};  // the process-wide kernel-launch counters at the last reset (a reset never clears them: another model,
                                    // or another thread's, keeps counting from its own base)
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;   // recycled events
};
Profiler* profiler_create() { return new Profiler(); }
void profiler_reset(Profiler* p) {
    for (auto& r : p->recs) {
        p->pool.push_back(r.e0);
        p->pool.push_back(r.e1);
    }
    p->recs.clear();
    for (auto& n : p->seen) n = 0;
    for (int f = 0; f < FAM_COUNT; ++f) p->launch_base[f] = kernel_launches(f);
}
int64_t profiler_kernel_launches(const Profiler* p, int family) {
    return (p && family >= 0 && family < FAM_COUNT) ? kernel_launches(family) - p->launch_base[family] : 0;
}
void profiler_destroy(Profiler* p) {
    if (!p) return;
    profiler_reset(p);
    for (auto e : p->pool) (void)hipEventDestroy(e);
    delete p;
}
void profiler_enable(Profiler* p, bool on) { p->enabled = on; }
void profiler_set_mask(Profiler* p, unsigned mask) { p->mask = mask ? mask : 0xFFFFFFFFu; }
bool profiler_enabled(const Profiler* p) { return p->enabled; }
static hipEvent_t prof_event(Profiler* p) {
    if (!p->pool.empty()) {
        hipEvent_t e = p->pool.back();
        p->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void profiler_set_stride(Profiler* p, int stride) { p->stride = stride < 1 ? 1 : stride; }
int64_t profiler_seen(const Profiler* p, int family) { return p->seen[family & 31]; }
int profiler_begin(Profiler* p, int family, double flops, double bytes, hipStream_t s) {
    if (!p || !p->enabled || !((p->mask >> family) & 1u)) return -1;
    if ((p->seen[family & 31]++ % p->stride) != 0) return -1;
    ProfRec r{family, flops, bytes, prof_event(p), prof_event(p)};
    (void)hipEventRecord(r.e0, s);
    p->recs.push_back(r);
    return (int)p->recs.size() - 1;
}
void profiler_end(Profiler* p, int token, hipStream_t s) {
    if (!p || token < 0 || token >= (int)p->recs.size()) return;
    (void)hipEventRecord(p->recs[token].e1, s);
}
int profiler_read(Profiler* p, int family, int64_t* launches, double* ms, double* flops, double* bytes) {
    int64_t n = 0;
    double t = 0, f = 0, by = 0;
    for (auto& r : p->recs) {
        if (r.family != family) continue;
        W2V2_HIP_CHECK(hipEventSynchronize(r.e1));
        float dt = 0.f;
        W2V2_HIP_CHECK(hipEventElapsedTime(&dt, r.e0, r.e1));
        t += dt; f += r.flops; by += r.bytes; ++n;
    }
    *launches = n; *ms = t; *flops = f; *bytes = by;
    return W2V2_OK;
}

}  // namespace w2v2

using namespace w2v2;

static void add_param(w2v2_model* m, const std::string& name, std::vector<int64_t> shape) {
    Param p;
    p.name = name;
    p.shape = shape;
    p.numel = 1;
    for (auto d : shape) p.numel *= d;
    m->index[name] = (int)m->params.size();
    m->params.push_back(p);
}

// Same inventory and order as gsoc-wav2vec2_amd/wav2vec2/variables.py (the
// reference's 213 variables for base CTC).
static void build_inventory(w2v2_model* m) {
    const w2v2_config& c = m->cfg;
    const int64_t H = c.hidden_size, F = c.intermediate_size;
    add_param(m, "masked_spec_embed", {H});
    int64_t cin = 1;
    for (int i = 0; i < c.num_conv_layers; ++i) {
        const std::string b = "feature_extractor/conv_layers/" + std::to_string(i);
        add_param(m, b + "/conv/kernel", {c.kernal_sizes[i], cin, c.filter_sizes[i]});
        if (c.conv_bias) add_param(m, b + "/conv/bias", {c.filter_sizes[i]});
        if (c.feature_extractor_norm_type == 1 || i == 0) {
            add_param(m, b + "/layer_norm/gamma", {c.filter_sizes[i]});
            add_param(m, b + "/layer_norm/beta", {c.filter_sizes[i]});
        }
        cin = c.filter_sizes[i];
    }
    add_param(m, "feature_projection/layer_norm/gamma", {cin});
    add_param(m, "feature_projection/layer_norm/beta", {cin});
    add_param(m, "feature_projection/projection/kernel", {cin, H});
    add_param(m, "feature_projection/projection/bias", {H});
    const int64_t K = c.num_conv_pos_embeddings, G = c.num_conv_pos_embedding_groups;
    add_param(m, "encoder/pos_conv_embed/conv/bias", {H});
    add_param(m, "encoder/pos_conv_embed/conv/weight_g", {K, 1, 1});
    add_param(m, "encoder/pos_conv_embed/conv/weight_v", {K, H / G, H});
    add_param(m, "encoder/layer_norm/gamma", {H});
    add_param(m, "encoder/layer_norm/beta", {H});
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string b = "encoder/layers/" + std::to_string(i);
        for (const char* p : {"q_proj", "k_proj", "v_proj", "out_proj"}) {
            add_param(m, b + "/attention/" + p + "/kernel", {H, H});
            add_param(m, b + "/attention/" + p + "/bias", {H});
        }
        add_param(m, b + "/layer_norm/gamma", {H});
        add_param(m, b + "/layer_norm/beta", {H});
        add_param(m, b + "/feed_forward/intermediate_dense/kernel", {H, F});
        add_param(m, b + "/feed_forward/intermediate_dense/bias", {F});
        add_param(m, b + "/feed_forward/output_dense/kernel", {F, H});
        add_param(m, b + "/feed_forward/output_dense/bias", {H});
        add_param(m, b + "/final_layer_norm/gamma", {H});
        add_param(m, b + "/final_layer_norm/beta", {H});
    }
    if (c.with_lm_head) {
        add_param(m, "lm_head/kernel", {H, c.vocab_size});
        add_param(m, "lm_head/bias", {c.vocab_size});
    }
}

static void free_planes(w2v2_model* m) {
    for (void* p : m->pl_allocs) (void)hipFree(p);
    m->pl_allocs.clear();
    m->conv48.clear();
    m->ln512_48 = m->attn_in48 = m->ctx48 = m->ffn_in48 = m->ffn48 = w2v2_model::PlaneBuf{};
    m->pl_fmt = -1;
    m->pl_B = 0;
    m->pl_L = 0;
}

static void free_workspace(w2v2_model* m) {
    free_planes(m);
    for (void* p : m->allocs) (void)hipFree(p);
    m->allocs.clear();
    for (void* p : m->sh_allocs) (void)hipFree(p);
    m->sh_allocs.clear();
    m->conv16.clear();
    m->hs16.clear();
    m->sh_ready = false;
    m->pos_pack16 = nullptr;       // lives in `allocs`
    m->acts.clear();
    m->conv.clear();
    m->conv_T.clear();
    m->hs.clear();
    m->ws_B = 0;
    m->ws_L = 0;
}

static int ws_alloc(w2v2_model* m, float** out, int64_t floats) {
    void* p = nullptr;
    W2V2_HIP_CHECK(hipMalloc(&p, (size_t)(floats > 0 ? floats : 1) * sizeof(float)));
    m->allocs.push_back(p);
    *out = reinterpret_cast<float*>(p);
    return W2V2_OK;
}

int w2v2_ensure_workspace(w2v2_model* m, int B, int64_t L) {
    if (m->ws_B == B && m->ws_L == L) return W2V2_OK;
    free_workspace(m);
    const w2v2_config& c = m->cfg;
    const int64_t H = c.hidden_size, F = c.intermediate_size;
    int64_t T = L;
    for (int i = 0; i < c.num_conv_layers; ++i) {
        T = 1 + (T - c.kernal_sizes[i]) / c.strides[i];
        float* p = nullptr;
        if (int e = ws_alloc(m, &p, (int64_t)B * T * c.filter_sizes[i])) return e;
        m->conv.push_back(p);
        m->conv_T.push_back((int)T);
        m->acts["conv" + std::to_string(i)] = Act{p, {B, T, c.filter_sizes[i]}};
    }
    const int64_t C = c.filter_sizes[c.num_conv_layers - 1];
    const int64_t BT = (int64_t)B * T;
    if (int e = ws_alloc(m, &m->conv0_ws, conv0_ws_floats(B, L, c.kernal_sizes[0], c.strides[0], c.filter_sizes[0]))) return e;
    if (int e = ws_alloc(m, &m->ln512, BT * C)) return e;
    if (int e = ws_alloc(m, &m->proj, BT * H)) return e;
    if (int e = ws_alloc(m, &m->posout, BT * H)) return e;
    m->acts["projection"] = Act{m->proj, {B, T, H}};
    for (int i = 0; i <= c.num_layers; ++i) {
        float* p = nullptr;
        if (c.attention_norm_type == 1 && i == 0) {
            p = m->posout;           // prenorm: encoder_in IS x + pos_conv(x)
        } else if (int e = ws_alloc(m, &p, BT * H)) {
            return e;
        }
        m->hs.push_back(p);
        m->acts[i == 0 ? std::string("encoder_in") : "layer" + std::to_string(i - 1)] = Act{p, {B, T, H}};
    }
    if (int e = ws_alloc(m, &m->qkv, BT * 3 * H)) return e;
    if (int e = ws_alloc(m, &m->ctx, BT * H)) return e;
    if (int e = ws_alloc(m, &m->t0, BT * H)) return e;
    if (int e = ws_alloc(m, &m->t1, BT * H)) return e;
    if (int e = ws_alloc(m, &m->t2, BT * H)) return e;
    if (int e = ws_alloc(m, &m->t3, BT * H)) return e;
    if (int e = ws_alloc(m, &m->ffn, BT * F)) return e;
    if (c.attention_norm_type == 1) {
        if (int e = ws_alloc(m, &m->enc_out, BT * H)) return e;
    } else {
        m->enc_out = m->hs[c.num_layers];
    }
    m->acts["encoder_out"] = Act{m->enc_out, {B, T, H}};
    float* fl = nullptr;
    if (int e = ws_alloc(m, &fl, B + 4)) return e;
    m->frame_len = reinterpret_cast<int32_t*>(fl);
    m->ws_B = B;
    m->ws_L = L;
    return W2V2_OK;
}

// ---- bf16 shadows for the inference forward in precision mode 1 ------------------------------------------
// w2v2_set_option(m, W2V2_OPT_BF16_SHADOWS, 0) turns them off (every GEMM then rounds its fp32 operands itself): same
// results bit for bit, used by the tests to prove exactly that.
bool w2v2_shadows_enabled(const w2v2_model* m) { return m->opt_shadows; }

// Precision mode 1 with shadows: a conv-stack output whose only consumer is the next layer's GEMM (reading the bf16 shadow)
// is written ONLY as bf16 -- 6.3 GB of fp32 stores per B = 32 x 246000 forward that nothing would read.  Stage taps of
// those tensors (w2v2_copy_activation) then report an error; w2v2_set_option(m, W2V2_OPT_KEEP_ACTIVATIONS, 1) keeps the fp32 copies.
bool w2v2_keep_activations(const w2v2_model* m) { return m->opt_keep_acts; }

#ifdef W2V2_TUNING
namespace w2v2 {
int tune_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
}  // namespace w2v2
#endif

// Whether conv-stack output i (0 .. NC-2) may be written ONLY as bf16 in the coming forward: group-norm mode with shadows,
// and layer i+1's GEMM is certain to take the bf16 shadow as its A operand (gemm_bf16.hip: K % 64 == 0 and 16-byte
// aligned rows / batch strides) -- otherwise that GEMM reads the fp32 tensor and it must exist.
bool w2v2_conv_out_bf16_only(const w2v2_model* m, int i, bool sh) {
    const w2v2_config& c = m->cfg;
    if (!sh || c.feature_extractor_norm_type == 1 || i + 1 >= c.num_conv_layers || w2v2_keep_activations(m)) return false;
    const int64_t cin = c.filter_sizes[i], K = (int64_t)c.kernal_sizes[i + 1] * cin, lda = (int64_t)c.strides[i + 1] * cin;
    const int64_t strideA = (int64_t)m->conv_T[i] * cin;
    return K % 64 == 0 && lda % 8 == 0 && strideA % 8 == 0;
}

// LayerNorm-mode extractor (robust / xlsr): conv i's LayerNorm + GELU output is written ONLY as bf16 when its one consumer, conv
// i + 1's GEMM, streams the shadow (same alignment conditions as above; the GEMM's own fp32 output is the LayerNorm's input and stays)
bool w2v2_conv_ln_bf16_only(const w2v2_model* m, int i, bool sh) {
    const w2v2_config& c = m->cfg;
    if (!sh || c.feature_extractor_norm_type != 1 || i + 1 >= c.num_conv_layers || w2v2_keep_activations(m)) return false;
    const int64_t cin = c.filter_sizes[i], K = (int64_t)c.kernal_sizes[i + 1] * cin, lda = (int64_t)c.strides[i + 1] * cin;
    const int64_t strideA = (int64_t)m->conv_T[i] * cin;
    return K % 64 == 0 && lda % 8 == 0 && strideA % 8 == 0;
}

static int sh_alloc(std::vector<void*>& pool, uint16_t** out, int64_t n) {
    void* p = nullptr;
    W2V2_HIP_CHECK(hipMalloc(&p, (size_t)(n > 0 ? n : 1) * sizeof(uint16_t)));
    pool.push_back(p);
    *out = reinterpret_cast<uint16_t*>(p);
    return W2V2_OK;
}

int w2v2_ensure_shadows(w2v2_model* m, int B, int T, hipStream_t s) {
    const w2v2_config& c = m->cfg;
    const int64_t H = c.hidden_size, F = c.intermediate_size, BT = (int64_t)B * T;
    if (!m->sh_ready) {
        for (int i = 0; i + 1 < c.num_conv_layers; ++i) {       // the last conv output feeds a LayerNorm, not a GEMM
            uint16_t* p = nullptr;
            if (int e = sh_alloc(m->sh_allocs, &p, (int64_t)B * m->conv_T[i] * c.filter_sizes[i])) return e;
            m->conv16.push_back(p);
        }
        if (int e = sh_alloc(m->sh_allocs, &m->ln512_16, BT * c.filter_sizes[c.num_conv_layers - 1])) return e;
        for (int i = 0; i <= c.num_layers; ++i) {
            uint16_t* p = nullptr;
            if (int e = sh_alloc(m->sh_allocs, &p, BT * H)) return e;
            m->hs16.push_back(p);
        }
        if (int e = sh_alloc(m->sh_allocs, &m->ctx16, BT * H)) return e;
        if (int e = sh_alloc(m->sh_allocs, &m->qkv16, BT * 3 * H)) return e;
        if (int e = sh_alloc(m->sh_allocs, &m->t0_16, BT * H)) return e;
        if (int e = sh_alloc(m->sh_allocs, &m->t2_16, BT * H)) return e;
        if (int e = sh_alloc(m->sh_allocs, &m->ffn16, BT * F)) return e;
        if (int e = sh_alloc(m->sh_allocs, &m->enc16, BT * H)) return e;
        m->sh_ready = true;
    }
    if (!m->w16_valid) {
        // (re)build the job table when it does not exist yet or the plain copies have become necessary (training started)
        const bool want_plain = m->train != nullptr;
        if (!m->shadow_jobs || (want_plain && !m->shadow_jobs_train)) {
            std::vector<ShadowJob> jobs;
            auto shadow = [&](const float* w, int K, int N) -> int {
                uint16_t*& dst = m->w16[w];
                if (!dst)
                    if (int e = sh_alloc(m->w16_allocs, &dst, (int64_t)K * N)) return e;
                uint16_t* plain = nullptr;
                if (want_plain && N % 64 == 0 && (K * (int64_t)N) % 4 == 0) {     // training: the backward's dX GEMM contracts over N
                    uint16_t*& dp = m->w16p[w];
                    if (!dp)
                        if (int e = sh_alloc(m->w16_allocs, &dp, (int64_t)K * N)) return e;
                    plain = dp;
                }
                for (int k0 = 0; k0 < K; k0 += 64)
                    for (int n0 = 0; n0 < N; n0 += 64) jobs.push_back(ShadowJob{w, dst, plain, K, N, k0, n0});
                return W2V2_OK;
            };
            for (int i = 1; i < c.num_conv_layers; ++i)
                if (int e = shadow(m->P("feature_extractor/conv_layers/" + std::to_string(i) + "/conv/kernel"),
                                   c.kernal_sizes[i] * c.filter_sizes[i - 1], c.filter_sizes[i]))
                    return e;
            if (int e = shadow(m->P("feature_projection/projection/kernel"), c.filter_sizes[c.num_conv_layers - 1], (int)H)) return e;
            for (int i = 0; i < c.num_layers; ++i) {
                const std::string b = "encoder/layers/" + std::to_string(i);
                if (int e = shadow(m->qkv_w[i], (int)H, 3 * (int)H)) return e;
                if (int e = shadow(m->P(b + "/attention/out_proj/kernel"), (int)H, (int)H)) return e;
                if (int e = shadow(m->P(b + "/feed_forward/intermediate_dense/kernel"), (int)H, (int)F)) return e;
                if (int e = shadow(m->P(b + "/feed_forward/output_dense/kernel"), (int)F, (int)H)) return e;
            }
            if (c.with_lm_head)
                if (int e = shadow(m->P("lm_head/kernel"), (int)H, c.vocab_size)) return e;
            if (m->shadow_jobs) W2V2_HIP_CHECK(hipFree(m->shadow_jobs));
            m->shadow_jobs = nullptr;
            W2V2_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m->shadow_jobs), jobs.size() * sizeof(ShadowJob)));
            W2V2_HIP_CHECK(hipMemcpy(m->shadow_jobs, jobs.data(), jobs.size() * sizeof(ShadowJob), hipMemcpyHostToDevice));
            m->shadow_njobs = (int)jobs.size();
            m->shadow_jobs_train = want_plain;
        }
        if (int e = launch_weight_shadows_multi(m->shadow_jobs, m->shadow_njobs, s)) return e;
        m->w16_valid = true;
    }
    return W2V2_OK;
}

// ---- precision modes 2 / 3: operand planes (gemm_split_sw.hip) ----------------------------------------------------------------
int w2v2_ensure_planes(w2v2_model* m, int B, int64_t L, int fmt) {
    if (m->pl_fmt == fmt && m->pl_B == B && m->pl_L == L) return W2V2_OK;
    free_planes(m);
    const w2v2_config& c = m->cfg;
    const int np = plane_count(fmt);
    auto alloc = [&](w2v2_model::PlaneBuf& b, int64_t elems) -> int {
        b.plane = (elems + 7) & ~(int64_t)7;                 // 16-byte aligned planes
        void* p = nullptr;
        W2V2_HIP_CHECK(hipMalloc(&p, (size_t)b.plane * np * sizeof(uint16_t)));
        m->pl_allocs.push_back(p);
        b.p = reinterpret_cast<uint16_t*>(p);
        return W2V2_OK;
    };
    const int NC = c.num_conv_layers;
    m->conv48.resize(NC > 1 ? NC - 1 : 0);
    for (int i = 0; i + 1 < NC; ++i)
        if (int e = alloc(m->conv48[i], (int64_t)B * m->conv_T[i] * c.filter_sizes[i])) return e;
    const int64_t BT = (int64_t)B * m->conv_T[NC - 1], H = c.hidden_size;
    if (int e = alloc(m->ln512_48, BT * c.filter_sizes[NC - 1])) return e;
    if (int e = alloc(m->attn_in48, BT * H)) return e;
    if (int e = alloc(m->ctx48, BT * H)) return e;
    if (int e = alloc(m->ffn_in48, BT * H)) return e;
    if (int e = alloc(m->ffn48, BT * c.intermediate_size)) return e;
    if (!m->range_flag) {
        W2V2_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m->range_flag), sizeof(int)));
        W2V2_HIP_CHECK(hipMemset(m->range_flag, 0, sizeof(int)));
    }
    m->pl_fmt = fmt;
    m->pl_B = B;
    m->pl_L = L;
    return W2V2_OK;
}

int w2v2_split_images(w2v2_model* m, const float* W, int K, int N, int fmt, hipStream_t s, const uint16_t** img, const float** out_scale) {
    W2V2_REQUIRE(W && (fmt == PF_BF16X3 || fmt == PF_F16X2), "split_images: bad argument");
    w2v2_model::SplitImages& e = m->wimg[fmt][W];
    const int64_t elems = (int64_t)plane_count(fmt) * K * N;
    if (!e.img || e.elems != elems) {
        if (e.img) (void)hipFree(e.img);
        e.img = nullptr;
        W2V2_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e.img), (size_t)elems * sizeof(uint16_t)));
        if (!e.scale_ws) W2V2_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e.scale_ws), 2 * sizeof(float)));
        e.elems = elems;
        e.epoch = 0;
    }
    if (e.epoch != m->w48_epoch) {
        if (int err = launch_split_weight_sw(W, e.img, K, N, fmt, e.scale_ws, s)) return err;
        e.epoch = m->w48_epoch;
    }
    *img = e.img;
    *out_scale = fmt == PF_F16X2 ? e.scale_ws + 1 : nullptr;
    return W2V2_OK;
}

bool w2v2_use_split_gemm(const w2v2_model* m, const float* A, int64_t lda, int64_t strideA, int64_t ldb, int M, int N, int K, int nbatch) {
    // (precision mode 3 falls back to this six-product kernel for the shapes / call sites its plane-fed kernel does not serve)
    if (m->precision < W2V2_PRECISION_BF16X3 || ldb != N || N % 256 != 0) return false;
    if (tune_int("W2V2_SPLIT_GEMM", 1) == 0) return false;      // (tools-only: tools/nll_drift_probe.py separates the GEMMs from the attention)
    // (below ~half a wave of 128 x 256 tiles the fp32 path's small tiles and split-K serve a single utterance better)
    const int64_t split_tiles = (int64_t)((M + 127) / 128) * (N / 256) * nbatch;
    return split_tiles >= 128 && gemm_split_supported(A, lda, strideA, M, N, K);
}

int w2v2_split_planes(w2v2_model* m, const float* W, int K, int N, hipStream_t s, const uint16_t** planes) {
    w2v2_model::SplitPlanes& e = m->w48[W];
    const int64_t need = 3 * (int64_t)K * N;
    if (!e.p || e.elems != need) {
        if (e.p) W2V2_HIP_CHECK(hipFree(e.p));
        e.p = nullptr;
        W2V2_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e.p), (size_t)need * sizeof(uint16_t)));
        e.elems = need;
        e.epoch = 0;
    }
    if (e.epoch != m->w48_epoch) {
        if (int err = launch_split_weight(W, e.p, K, N, s)) return err;
        e.epoch = m->w48_epoch;
    }
    *planes = e.p;
    return W2V2_OK;
}

bool w2v2_pos_conv_bf16_ok(const w2v2_model* m) {
    const int cg = m->cfg.hidden_size / m->cfg.num_conv_pos_embedding_groups;
    return m->precision == 1 && cg % 8 == 0 && cg <= 64 && (m->cfg.num_conv_pos_embeddings * cg) % 64 == 0;
}

int w2v2_ensure_pos16(w2v2_model* m, int B, int T, hipStream_t s) {
    const w2v2_config& c = m->cfg;
    const int H = c.hidden_size, K = c.num_conv_pos_embeddings, G = c.num_conv_pos_embedding_groups, cg = H / G;
    if (!m->pos_w16) W2V2_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m->pos_w16), (size_t)K * cg * H * sizeof(uint16_t)));
    if (!m->pos16_valid) {
        if (int e = launch_pos_conv_weight_shadow(m->pos_wg, m->pos_w16, K, cg, G, s)) return e;
        m->pos16_valid = true;
    }
    if (!m->pos_pack16) {
        float* p = nullptr;
        if (int e = ws_alloc(m, &p, (pos_conv_bf16_pack_elems(B, T, H, K) + 1) / 2 + 4)) return e;
        m->pos_pack16 = reinterpret_cast<uint16_t*>(p);
    }
    return W2V2_OK;
}

extern "C" {

const char* w2v2_last_error(void) { return g_err; }
int w2v2_release_scratch(void) { return w2v2::stream_scratch_release(); }
const char* w2v2_version(void) { return "w2v2-gfx950 0.1 (fp32 MFMA path; bf16-operand and bf16x3-split precision modes)"; }

int w2v2_create(const w2v2_config* cfg, w2v2_model** out) {
    W2V2_REQUIRE(cfg && out, "create: null argument");
    const w2v2_config& c = *cfg;
    W2V2_REQUIRE(c.num_conv_layers > 0 && c.num_conv_layers <= W2V2_MAX_CONV_LAYERS, "create: num_conv_layers=%d", c.num_conv_layers);
    W2V2_REQUIRE(c.hidden_size > 0 && c.num_heads > 0 && c.hidden_size % c.num_heads == 0,
                 "create: hidden_size %d is not a multiple of num_heads %d", c.hidden_size, c.num_heads);
    W2V2_REQUIRE(c.num_conv_pos_embedding_groups > 0 && c.hidden_size % c.num_conv_pos_embedding_groups == 0,
                 "create: hidden_size %d is not a multiple of the positional conv groups", c.hidden_size);
    W2V2_REQUIRE(c.feature_extractor_norm_type == 0 || c.feature_extractor_norm_type == 1, "create: bad conv norm type");
    W2V2_REQUIRE(c.attention_norm_type == 0 || c.attention_norm_type == 1, "create: bad attention norm type");
    W2V2_REQUIRE(c.num_layers >= 1 && c.intermediate_size > 0 && c.vocab_size > 0, "create: bad transformer sizes");
    for (int i = 0; i < c.num_conv_layers; ++i) {
        W2V2_REQUIRE(c.filter_sizes[i] > 0 && c.kernal_sizes[i] > 0 && c.strides[i] > 0, "create: bad conv layer %d", i);
        W2V2_REQUIRE(c.filter_sizes[i] % 4 == 0, "create: conv filters must be a multiple of 4");
    }
    w2v2_model* m = new w2v2_model();
    m->cfg = c;
    build_inventory(m);
    for (auto& p : m->params) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p.dev), (size_t)p.numel * sizeof(float));
        if (e == hipSuccess) e = hipMemset(p.dev, 0, (size_t)p.numel * sizeof(float));
        if (e != hipSuccess) {
            set_error("create: allocating `%s` failed: %s", p.name.c_str(), hipGetErrorString(e));
            w2v2_destroy(m);
            return W2V2_EHIP;
        }
    }
    m->prof = profiler_create();
    *out = m;
    return W2V2_OK;
}

void w2v2_destroy(w2v2_model* m) {
    if (!m) return;
    w2v2_comm_free(m);
    w2v2_train_destroy(m);
    free_workspace(m);
    for (auto& p : m->params)
        if (p.dev) (void)hipFree(p.dev);
    if (m->pos_wg) (void)hipFree(m->pos_wg);
    for (auto p : m->qkv_w) (void)hipFree(p);
    for (auto p : m->qkv_b) (void)hipFree(p);
    for (void* p : m->w16_allocs) (void)hipFree(p);
    for (auto& kv : m->w48)
        if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto& tab : m->wimg)
        for (auto& kv : tab) {
            if (kv.second.img) (void)hipFree(kv.second.img);
            if (kv.second.scale_ws) (void)hipFree(kv.second.scale_ws);
        }
    if (m->range_flag) (void)hipFree(m->range_flag);
    if (m->pos_w16) (void)hipFree(m->pos_w16);
    if (m->shadow_jobs) (void)hipFree(m->shadow_jobs);
    profiler_destroy(m->prof);
    delete m;
}

int w2v2_num_params(const w2v2_model* m) { return m ? (int)m->params.size() : 0; }

int w2v2_param_info(const w2v2_model* m, int index, const char** name, int64_t shape[4], int* rank) {
    W2V2_REQUIRE(m && index >= 0 && index < (int)m->params.size(), "param_info: bad index %d", index);
    const Param& p = m->params[index];
    if (name) *name = p.name.c_str();
    if (rank) *rank = (int)p.shape.size();
    if (shape)
        for (size_t i = 0; i < 4; ++i) shape[i] = i < p.shape.size() ? p.shape[i] : 1;
    return W2V2_OK;
}

int w2v2_set_param(w2v2_model* m, const char* name, const float* host_src, const int64_t* shape, int rank) {
    W2V2_REQUIRE(m && name && host_src && shape, "set_param: null argument");
    auto it = m->index.find(name);
    if (it == m->index.end()) {
        set_error("set_param: unknown variable `%s`", name);
        return W2V2_ENOTFOUND;
    }
    Param& p = m->params[it->second];
    bool ok = rank == (int)p.shape.size();
    for (int i = 0; ok && i < rank; ++i) ok = shape[i] == p.shape[i];
    W2V2_REQUIRE(ok, "set_param: shape mismatch for `%s`", name);
    W2V2_HIP_CHECK(hipMemcpy(p.dev, host_src, (size_t)p.numel * sizeof(float), hipMemcpyHostToDevice));
    p.set = true;
    m->finalized = false;
    w2v2_train_invalidate(m);
    return W2V2_OK;
}

int w2v2_get_param(w2v2_model* m, const char* name, float* host_dst, int64_t numel) {
    W2V2_REQUIRE(m && name && host_dst, "get_param: null argument");
    auto it = m->index.find(name);
    if (it == m->index.end()) {
        set_error("get_param: unknown variable `%s`", name);
        return W2V2_ENOTFOUND;
    }
    Param& p = m->params[it->second];
    W2V2_REQUIRE(numel == p.numel, "get_param: `%s` has %lld elements, caller asked for %lld", name,
                 (long long)p.numel, (long long)numel);
    W2V2_HIP_CHECK(hipDeviceSynchronize());
    W2V2_HIP_CHECK(hipMemcpy(host_dst, p.dev, (size_t)p.numel * sizeof(float), hipMemcpyDeviceToHost));
    return W2V2_OK;
}

int w2v2_finalize(w2v2_model* m, void* stream) {
    W2V2_REQUIRE(m, "finalize: null model");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const w2v2_config& c = m->cfg;
    const int64_t H = c.hidden_size;
    const int K = c.num_conv_pos_embeddings, G = c.num_conv_pos_embedding_groups, cg = (int)(H / G);
    if (!m->pos_wg) W2V2_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m->pos_wg), (size_t)K * cg * H * sizeof(float)));
    if (int e = launch_weight_norm_regroup(m->prof, m->P("encoder/pos_conv_embed/conv/weight_v"),
                                           m->P("encoder/pos_conv_embed/conv/weight_g"), m->pos_wg, K, cg,
                                           (int)H, G, s))
        return e;
    // packed q|k|v projection: one (H, 3H) GEMM per layer instead of three (H, H)
    if (m->qkv_w.empty()) {
        m->qkv_w.resize(c.num_layers, nullptr);
        m->qkv_b.resize(c.num_layers, nullptr);
        for (int i = 0; i < c.num_layers; ++i) {
            W2V2_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m->qkv_w[i]), (size_t)H * 3 * H * sizeof(float)));
            W2V2_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m->qkv_b[i]), (size_t)3 * H * sizeof(float)));
        }
    }
    if (H % 4 == 0 && c.num_layers > 0) {       // all layers in one launch (24 per launch): this runs after every optimizer step
        std::vector<const float*> wl(3 * (size_t)c.num_layers), bl(3 * (size_t)c.num_layers);
        for (int i = 0; i < c.num_layers; ++i) {
            const std::string b = "encoder/layers/" + std::to_string(i) + "/attention/";
            const char* names[3] = {"q_proj", "k_proj", "v_proj"};
            for (int j = 0; j < 3; ++j) {
                wl[3 * i + j] = m->P(b + names[j] + "/kernel");
                bl[3 * i + j] = m->P(b + names[j] + "/bias");
            }
        }
        if (int e = launch_qkv_pack_layers(m->qkv_w.data(), m->qkv_b.data(), wl.data(), bl.data(), c.num_layers, H, s)) return e;
    }
    for (int i = 0; i < c.num_layers && H % 4 != 0; ++i) {
        const std::string b = "encoder/layers/" + std::to_string(i) + "/attention/";
        const char* names[3] = {"q_proj", "k_proj", "v_proj"};
        const float* wj[3];
        const float* bj[3];
        for (int j = 0; j < 3; ++j) {
            wj[j] = m->P(b + names[j] + "/kernel");
            bj[j] = m->P(b + names[j] + "/bias");
        }
        {
            for (int j = 0; j < 3; ++j) {
                W2V2_HIP_CHECK(hipMemcpy2DAsync(m->qkv_w[i] + j * H, (size_t)3 * H * sizeof(float), wj[j], (size_t)H * sizeof(float),
                                                (size_t)H * sizeof(float), (size_t)H, hipMemcpyDeviceToDevice, s));
                W2V2_HIP_CHECK(hipMemcpyAsync(m->qkv_b[i] + j * H, bj[j], (size_t)H * sizeof(float), hipMemcpyDeviceToDevice, s));
            }
        }
    }
    m->finalized = true;
    m->w16_valid = false;      // the bf16 weight shadows (if any) follow the variables
    ++m->w48_epoch;
    m->pos16_valid = false;
    return W2V2_OK;
}

int64_t w2v2_num_frames(const w2v2_model* m, int64_t n) {
    if (!m) return -1;
    for (int i = 0; i < m->cfg.num_conv_layers; ++i) {
        if (n < m->cfg.kernal_sizes[i]) return 0;
        n = 1 + (n - m->cfg.kernal_sizes[i]) / m->cfg.strides[i];
    }
    return n;
}

int w2v2_set_precision(w2v2_model* m, int32_t mode) {
    W2V2_REQUIRE(m, "set_precision: null model");
    W2V2_REQUIRE(mode >= W2V2_PRECISION_FP32 && mode <= W2V2_PRECISION_F16X2, "set_precision: unknown mode %d", mode);
    m->precision = mode;
    return W2V2_OK;
}
int w2v2_get_precision(const w2v2_model* m) { return m ? m->precision : W2V2_EINVAL; }

int w2v2_set_option(w2v2_model* m, int32_t option, int32_t value) {
    W2V2_REQUIRE(m, "set_option: null model");
    switch (option) {
        case W2V2_OPT_BF16_SHADOWS: m->opt_shadows = value != 0; return W2V2_OK;
        case W2V2_OPT_KEEP_ACTIVATIONS: m->opt_keep_acts = value != 0; return W2V2_OK;
        case W2V2_OPT_SPLIT_PLANES: m->opt_planes = value != 0; return W2V2_OK;
        case W2V2_OPT_WGRAD_STREAM: m->opt_wgrad_stream = value != 0; return W2V2_OK;
        case W2V2_OPT_DEFER_FOLDS: m->opt_defer_folds = value != 0; return W2V2_OK;
        default: set_error("set_option: unknown option %d", option); return W2V2_EINVAL;
    }
}
int w2v2_get_option(const w2v2_model* m, int32_t option) {
    if (!m) return W2V2_EINVAL;
    switch (option) {
        case W2V2_OPT_BF16_SHADOWS: return m->opt_shadows ? 1 : 0;
        case W2V2_OPT_KEEP_ACTIVATIONS: return m->opt_keep_acts ? 1 : 0;
        case W2V2_OPT_SPLIT_PLANES: return m->opt_planes ? 1 : 0;
        case W2V2_OPT_WGRAD_STREAM: return m->opt_wgrad_stream ? 1 : 0;
        case W2V2_OPT_DEFER_FOLDS: return m->opt_defer_folds ? 1 : 0;
        default: return W2V2_EINVAL;
    }
}

int w2v2_range_overflow(w2v2_model* m, int32_t* flag, void* stream) {
    W2V2_REQUIRE(m && flag, "range_overflow: null argument");
    *flag = 0;
    if (!m->range_flag) return W2V2_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int v = 0;
    W2V2_HIP_CHECK(hipMemcpyAsync(&v, m->range_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    W2V2_HIP_CHECK(hipStreamSynchronize(s));
    if (v) W2V2_HIP_CHECK(hipMemsetAsync(m->range_flag, 0, sizeof(int), s));
    *flag = v != 0;
    return W2V2_OK;
}

int w2v2_forward(w2v2_model* m, const float* wave, int32_t B, int64_t L, const int32_t* mask,
                 float* out, void* stream) {
    W2V2_REQUIRE(m && wave && out, "forward: null argument");
    W2V2_REQUIRE(B > 0 && L > 0, "forward: bad batch shape (%d, %lld)", B, (long long)L);
    if (!m->finalized) {
        set_error("forward: call w2v2_finalize after setting the variables");
        return W2V2_ESTATE;
    }
    const w2v2_config& c = m->cfg;
    PrecisionScope precision(m->precision);
    const int64_t Tll = w2v2_num_frames(m, L);
    W2V2_REQUIRE(Tll >= 1, "forward: %lld samples are shorter than the conv stack's receptive field", (long long)L);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (int e = w2v2_ensure_workspace(m, B, L)) return e;
    Profiler* pf = m->prof;
    const int T = (int)Tll;
    const int H = c.hidden_size, F = c.intermediate_size;
    const int64_t BT = (int64_t)B * T;
    const int act = c.is_gelu_approx ? 2 : 1;
    // element-wise kernels in precision mode 1 evaluate exact GELU through the 5-term erf the bf16 GEMM epilogue uses (act 3)
    const int act_ew = (act == 1 && m->precision == 1) ? 3 : act;
    const bool layer_mode = c.feature_extractor_norm_type == 1;
    const bool prenorm = c.attention_norm_type == 1;
    const float eps = c.layer_norm_eps;
    auto fe = [&](int i, const char* leaf) { return m->P("feature_extractor/conv_layers/" + std::to_string(i) + leaf); };

    // Precision mode 1 with bf16 shadows: every producer of a GEMM operand also writes its nearest-even bf16 copy, the
    // GEMMs stream those (2 bytes per element, no conversion) and the weights come from (N, K) bf16 shadows.  `sh`
    // false = plain pointers everywhere: the GEMMs then round their fp32 operands themselves, with identical results.
    const bool sh = m->precision == 1 && w2v2_shadows_enabled(m);
    if (sh)
        if (int e = w2v2_ensure_shadows(m, B, T, s)) return e;
    const bool attn16 = sh && attention_bf16_supported(H / c.num_heads);
    auto W16 = [&](const float* w) -> const uint16_t* { return sh ? m->w16[w] : nullptr; };
    // Precision mode 2: fp32 operands, each an exact sum of three bf16 terms, six MFMA products (gemm_split.hip).  The
    // weight planes are built on first use; shapes the split kernel does not take (lm_head: N = 32) stay on the fp32 MFMA.
    // Precision modes 2 / 3 with operand planes (W2V2_OPT_SPLIT_PLANES, default): every producer of a GEMM operand writes its planes
    // -- three bf16 terms (bf16x3) or two fp16 terms (f16x2) per element -- and the GEMM streams them (gemm_split_sw.hip).  Whether a
    // call site does is decided here from its shape, so that the producer knows: whole 256-column tiles, K % 64 == 0, 16-byte aligned
    // rows, and enough tiles to fill the chip (below that the fp32 path's small tiles serve a single utterance better).
    using PlaneBuf = w2v2_model::PlaneBuf;
    const int NC = c.num_conv_layers;
    const bool pm = m->precision >= W2V2_PRECISION_BF16X3 && m->opt_planes;
    const int fmt = m->precision == W2V2_PRECISION_F16X2 ? PF_F16X2 : PF_BF16X3;
    const bool keep = w2v2_keep_activations(m);
    auto site = [&](int64_t M_, int N_, int K_, int nb, int64_t lda_, int64_t sA_) {
        return pm && N_ % 256 == 0 && K_ % 64 == 0 && lda_ % 8 == 0 && sA_ % 8 == 0 && 128 * lda_ < (1 << 29) && ((M_ + 127) / 128) * (N_ / 256) * nb >= 128;
    };
    std::vector<char> cp(NC + 1, 0);             // cp[i]: conv layer i's GEMM streams the planes of conv output i - 1
    for (int i = 1; i < NC; ++i)
        cp[i] = site(m->conv_T[i], c.filter_sizes[i], c.kernal_sizes[i] * c.filter_sizes[i - 1], B, (int64_t)c.strides[i] * c.filter_sizes[i - 1],
                     (int64_t)m->conv_T[i - 1] * c.filter_sizes[i - 1]) && c.filter_sizes[i - 1] % 4 == 0;
    const int C512 = c.filter_sizes[NC - 1];
    const bool p_proj = site(BT, H, C512, 1, C512, 0) && C512 % 4 == 0, p_qkv = site(BT, 3 * H, H, 1, H, 0) && H % 4 == 0, p_out = site(BT, H, H, 1, H, 0),
               p_f1 = site(BT, F, H, 1, H, 0), p_f2 = site(BT, H, F, 1, F, 0) && F % 8 == 0;
    bool any_planes = p_proj || p_qkv || p_out || p_f1 || p_f2;
    for (int i = 1; i < NC; ++i) any_planes = any_planes || cp[i];
    if (any_planes)
        if (int e = w2v2_ensure_planes(m, B, L, fmt)) return e;
    auto PO = [&](const PlaneBuf& b) {
        PlaneOut o;
        o.p = b.p; o.plane = b.plane; o.fmt = fmt; o.range_flag = m->range_flag;
        return o;
    };
    // Apl: the planes of A (the call site was decided above), Cpl: where the planes of the result go, need_f32: the fp32 result is
    // wanted as well (a tap under W2V2_OPT_KEEP_ACTIVATIONS) -- the plane epilogue writes one or the other, so it is then split off C.
    auto gemm = [&](const float* A, const uint16_t* A16, int64_t lda, int64_t strideA, const float* Bw, int64_t ldb, float* Cc,
                    uint16_t* C16, int64_t ldc, int64_t strideC, const float* bias, const float* res, int M, int N, int K,
                    int nbatch, int act_, const PlaneBuf* Apl = nullptr, const PlaneBuf* Cpl = nullptr, bool need_f32 = true) -> int {
        auto split_out = [&]() -> int {
            W2V2_REQUIRE(Cc && ldc == N && (nbatch == 1 || strideC == (int64_t)M * N), "forward: plane output of a strided result");
            return launch_split_planes(Cc, Cpl->p, Cpl->plane, (int64_t)nbatch * M * N, fmt, m->range_flag, s);
        };
        if (Apl) {
            const uint16_t* img = nullptr;
            const float* sc = nullptr;
            if (int e = w2v2_split_images(m, Bw, K, N, fmt, s, &img, &sc)) return e;
            const bool planes_only = Cpl && !need_f32;
            if (int e = launch_gemm_split_sw(pf, fmt, Apl->p, Apl->plane, lda, strideA, img, sc, planes_only ? nullptr : Cc, planes_only ? Cpl->p : nullptr,
                                             planes_only ? Cpl->plane : 0, ldc, strideC, bias, res, M, N, K, nbatch, act_, m->range_flag, s))
                return e;
            return (Cpl && !planes_only) ? split_out() : W2V2_OK;
        }
        int e;
        if (w2v2_use_split_gemm(m, A, lda, strideA, ldb, M, N, K, nbatch)) {
            const uint16_t* planes = nullptr;
            if (int e2 = w2v2_split_planes(m, Bw, K, N, s, &planes)) return e2;
            e = launch_gemm_split(pf, A, lda, strideA, planes, Cc, ldc, strideC, bias, res, M, N, K, nbatch, act_, s);
        } else if (!sh) {
            e = launch_gemm(pf, A, lda, strideA, Bw, ldb, Cc, ldc, strideC, bias, res, M, N, K, nbatch, act_, s);
        } else {
            GemmShadows x;
            x.A16 = A16; x.B16 = W16(Bw); x.C16 = C16; x.ldb16 = K;
            e = launch_gemm_bf16_x(pf, A, lda, strideA, Bw, ldb, 0, Cc, ldc, strideC, bias, res, M, N, K, nbatch, act_, x, s);
        }
        if (e) return e;
        return Cpl ? split_out() : W2V2_OK;
    };
    const bool ffn_sh_only = sh && F % 64 == 0;      // the output dense can then always take the shadow (K = F)

    // ---- feature extractor (feature_extractor.py:54-59) ----
    // group-norm mode with shadows: conv0 .. conv(NC-2) feed only the next layer's GEMM; where that GEMM reads the bf16 shadow
    // the fp32 copy is not written at all (w2v2_conv_out_bf16_only)
    m->acts_skipped.clear();
    // (planes: a conv output whose one reader streams its planes is written ONLY as planes, like the bf16-only outputs of mode 1)
    auto planes_only_out = [&](int i) { return i + 1 < NC && cp[i + 1] && !keep; };
    // the fused plane output needs conv0's 16-byte-store kernel (K = 10, stride 5); other geometries write fp32 and split it
    const bool fused = NC > 1 && cp[1] && c.kernal_sizes[0] == 10 && c.strides[0] == 5 && 256 % (c.filter_sizes[0] / 4) == 0 && c.filter_sizes[0] / 4 <= 256;
    // "convI" is not readable back exactly when the code below hands its producer a null fp32 destination (this used to be a
    // second, looser predicate): conv0 fused into planes; a plane-fed GEMM whose output also goes out as planes only; a LayerNorm-mode
    // layer whose LN + GELU pass writes bf16 / planes only (conv[i] then holds the pre-norm values); the bf16-only outputs of mode 1
    auto f32_skipped = [&](int i) {
        if (w2v2_conv_out_bf16_only(m, i, sh) || w2v2_conv_ln_bf16_only(m, i, sh)) return true;
        if (i == 0) return NC > 1 && cp[1] && !keep && fused;
        if (layer_mode) return planes_only_out(i);
        return cp[i] && i + 1 < NC && cp[i + 1] && !keep;
    };
    for (int i = 0; i + 1 < NC; ++i)
        if (f32_skipped(i)) m->acts_skipped.push_back("conv" + std::to_string(i));
    {
        const PlaneOut po = fused ? PO(m->conv48[0]) : PlaneOut{};
        const bool f32_too = !(NC > 1 && cp[1]) || keep || !fused;
        if (int e = launch_conv0_x(pf, wave, fe(0, "/conv/kernel"), c.conv_bias ? fe(0, "/conv/bias") : nullptr,
                                   fe(0, "/layer_norm/gamma"), fe(0, "/layer_norm/beta"),
                                   (w2v2_conv_out_bf16_only(m, 0, sh) || w2v2_conv_ln_bf16_only(m, 0, sh) || !f32_too) ? nullptr : m->conv[0],
                                   sh ? m->conv16[0] : nullptr, m->conv0_ws, B, L, c.kernal_sizes[0], c.strides[0],
                                   c.filter_sizes[0], 1e-5f, layer_mode ? 2 : 0, act_ew, s, fused ? &po : nullptr))      // (layer mode: conv + LayerNorm + GELU in one pass)
            return e;
        if (NC > 1 && cp[1] && !fused)
            if (int e = launch_split_planes(m->conv[0], m->conv48[0].p, m->conv48[0].plane, (int64_t)B * m->conv_T[0] * c.filter_sizes[0], fmt, m->range_flag, s))
                return e;
    }
    for (int i = 1; i < NC; ++i) {
        const int cin = c.filter_sizes[i - 1], cout = c.filter_sizes[i];
        const int Tin = m->conv_T[i - 1], Tout = m->conv_T[i];
        uint16_t* o16 = (sh && i + 1 < NC) ? m->conv16[i] : nullptr;     // the last conv output feeds a LayerNorm
        // strided Conv1D == GEMM over an overlapping window view: lda = stride * C_in < K * C_in
        // planes of this layer's output for the next layer's GEMM: from the GEMM epilogue (group-norm mode: bias + GELU there) or from
        // the LayerNorm + GELU pass behind it (layer-norm mode: the GEMM output is that pass's fp32 input)
        const bool out_pl = i + 1 < NC && cp[i + 1];
        const PlaneBuf* opl = out_pl ? &m->conv48[i] : nullptr;
        if (int e = gemm(m->conv[i - 1], sh ? m->conv16[i - 1] : nullptr, (int64_t)c.strides[i] * cin, (int64_t)Tin * cin,
                         fe(i, "/conv/kernel"), cout, w2v2_conv_out_bf16_only(m, i, sh) ? nullptr : m->conv[i], layer_mode ? nullptr : o16, cout,
                         (int64_t)Tout * cout, c.conv_bias ? fe(i, "/conv/bias") : nullptr, nullptr, Tout, cout, c.kernal_sizes[i] * cin, B,
                         layer_mode ? 0 : act, cp[i] ? &m->conv48[i - 1] : nullptr, layer_mode ? nullptr : opl, keep))
            return e;
        if (layer_mode) {
            const PlaneOut po = out_pl ? PO(*opl) : PlaneOut{};
            if (int e = launch_layer_norm_x(pf, m->conv[i], (w2v2_conv_ln_bf16_only(m, i, sh) || planes_only_out(i)) ? nullptr : m->conv[i], fe(i, "/layer_norm/gamma"),
                                            fe(i, "/layer_norm/beta"), (int64_t)B * Tout, cout, 1e-5f, act_ew, o16, s, out_pl ? &po : nullptr))
                return e;
        }
    }
    // ---- feature projection (feature_extractor.py:92-95) ----
    const int C = c.filter_sizes[NC - 1];
    {
        const PlaneOut po = p_proj ? PO(m->ln512_48) : PlaneOut{};
        if (int e = launch_layer_norm_x(pf, m->conv[NC - 1], (p_proj && !keep) ? nullptr : m->ln512, m->P("feature_projection/layer_norm/gamma"),
                                        m->P("feature_projection/layer_norm/beta"), BT, C, eps, 0, sh ? m->ln512_16 : nullptr, s, p_proj ? &po : nullptr))
            return e;
    }
    if (int e = gemm(m->ln512, sh ? m->ln512_16 : nullptr, C, 0, m->P("feature_projection/projection/kernel"), H, m->proj, nullptr,
                     H, 0, m->P("feature_projection/projection/bias"), nullptr, (int)BT, H, C, 1, 0, p_proj ? &m->ln512_48 : nullptr))
        return e;
    // ---- encoder (encoder.py:251-276) ----
    const int32_t* flen = nullptr;
    if (mask) {
        if (int e = launch_frame_lengths(pf, mask, m->frame_len, B, L, c.kernal_sizes, c.strides, c.num_conv_layers, s)) return e;
        flen = m->frame_len;
    }
    if (w2v2_pos_conv_bf16_ok(m)) {      // precision mode 1: one batched bf16 GEMM over (sample, group); m->t0 is free here
        if (int e = w2v2_ensure_pos16(m, B, T, s)) return e;
        if (int e = launch_pos_conv_bf16(pf, m->proj, m->pos_w16, m->P("encoder/pos_conv_embed/conv/bias"), flen, m->posout, nullptr,
                                         m->pos_pack16, m->t0, B, T, H, c.num_conv_pos_embeddings, c.num_conv_pos_embedding_groups,
                                         act, c.num_conv_pos_embeddings / 2, 1, s))
            return e;
    } else if (int e = launch_pos_conv(pf, m->proj, m->pos_wg, m->P("encoder/pos_conv_embed/conv/bias"), flen, m->posout, B, T,
                                       H, c.num_conv_pos_embeddings, c.num_conv_pos_embedding_groups, act, s)) {
        return e;
    }
    const PlaneOut po_attn = p_qkv ? PO(m->attn_in48) : PlaneOut{}, po_ctx = p_out ? PO(m->ctx48) : PlaneOut{},
                   po_ffn_in = p_f1 ? PO(m->ffn_in48) : PlaneOut{};
    if (!prenorm)
        if (int e = launch_layer_norm_x(pf, m->posout, m->hs[0], m->P("encoder/layer_norm/gamma"),
                                        m->P("encoder/layer_norm/beta"), BT, H, eps, 0, sh ? m->hs16[0] : nullptr, s, p_qkv ? &po_attn : nullptr))
            return e;
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string b = "encoder/layers/" + std::to_string(i);
        const float* x = m->hs[i];
        const float* attn_in = x;
        const uint16_t* attn_in16 = sh ? m->hs16[i] : nullptr;       // postnorm: the previous LayerNorm wrote it
        // (prenorm, bf16 shadows: the two in-layer LayerNorm outputs feed one GEMM each; when that GEMM is certain to stream the shadow -- K = H a
        //  multiple of 64 and a weight shadow, as for the attention output below -- the fp32 copy is not written: 2 x 98 MB per layer at 16 x 480000)
        auto shadow_certain = [&](const float* w) {
            const auto it = m->w16.find(w);
            return sh && !keep && H % 64 == 0 && it != m->w16.end() && it->second != nullptr;
        };
        if (prenorm) {
            const bool a16_only = shadow_certain(m->qkv_w[i]);
            if (int e = launch_layer_norm_x(pf, x, ((p_qkv && !keep) || a16_only) ? nullptr : m->t0, m->P(b + "/layer_norm/gamma"), m->P(b + "/layer_norm/beta"), BT, H, eps, 0,
                                            sh ? m->t0_16 : nullptr, s, p_qkv ? &po_attn : nullptr))
                return e;
            attn_in = m->t0;
            attn_in16 = sh ? m->t0_16 : nullptr;
        }
        // the bf16 attention kernels read q | k | v only as bf16: the projection then writes just that shadow
        if (int e = gemm(attn_in, attn_in16, H, 0, m->qkv_w[i], 3 * H, attn16 ? nullptr : m->qkv, attn16 ? m->qkv16 : nullptr, 3 * H, 0, m->qkv_b[i],
                         nullptr, (int)BT, 3 * H, H, 1, 0, p_qkv ? &m->attn_in48 : nullptr))
            return e;
        // (the attention output's one reader is the out-projection GEMM; when that is certain to stream the bf16 shadow -- K = H a
        //  multiple of 64, rows 16-byte aligned: gemm_bf16.hip -- the fp32 copy is not written: 75 MB per layer at B = 32)
        const auto wo16 = m->w16.find(m->P(b + "/attention/out_proj/kernel"));
        const bool ctx16_only = attn16 && H % 64 == 0 && wo16 != m->w16.end() && wo16->second != nullptr;
        // (planes: the split attention kernel writes the planes of ctx itself; any other attention kernel leaves fp32 to be split)
        const bool ctx_fused = p_out && attention_split_supported(H / c.num_heads) && H % 4 == 0 && tune_int("W2V2_SPLIT_ATTN", 1) != 0;
        PlaneOut po_flag;                                    // (no planes wanted: the split attention still reports f16x2 saturation)
        po_flag.range_flag = m->range_flag;
        if (int e = launch_attention_x(pf, attn16 ? nullptr : m->qkv, attn16 ? m->qkv16 : nullptr, flen, (ctx16_only || (ctx_fused && !keep)) ? nullptr : m->ctx, B, T, H,
                                       c.num_heads, attn16 ? m->ctx16 : nullptr, s, ctx_fused ? &po_ctx : (pm ? &po_flag : nullptr)))
            return e;
        if (p_out && !ctx_fused)
            if (int e = launch_split_planes(m->ctx, m->ctx48.p, m->ctx48.plane, BT * H, fmt, m->range_flag, s)) return e;
        // out projection + residual (encoder.py:31,117-119)
        if (int e = gemm(m->ctx, attn16 ? m->ctx16 : nullptr, H, 0, m->P(b + "/attention/out_proj/kernel"), H, m->t1, nullptr, H, 0,
                         m->P(b + "/attention/out_proj/bias"), x, (int)BT, H, H, 1, 0, p_out ? &m->ctx48 : nullptr))
            return e;
        const float* ffn_res = m->t1;
        if (!prenorm) {
            if (int e = launch_layer_norm_x(pf, m->t1, m->t2, m->P(b + "/layer_norm/gamma"), m->P(b + "/layer_norm/beta"), BT, H, eps, 0,
                                            sh ? m->t2_16 : nullptr, s, p_f1 ? &po_ffn_in : nullptr))
                return e;
            ffn_res = m->t2;
        } else {
            const bool t2_16_only = shadow_certain(m->P(b + "/feed_forward/intermediate_dense/kernel"));
            if (int e = launch_layer_norm_x(pf, m->t1, ((p_f1 && !keep) || t2_16_only) ? nullptr : m->t2, m->P(b + "/final_layer_norm/gamma"), m->P(b + "/final_layer_norm/beta"), BT, H,
                                            eps, 0, sh ? m->t2_16 : nullptr, s, p_f1 ? &po_ffn_in : nullptr))
                return e;
        }
        // the FFN intermediate has one consumer: with shadows only its bf16 form is written (302 MB of fp32 stores saved)
        if (int e = gemm(m->t2, sh ? m->t2_16 : nullptr, H, 0, m->P(b + "/feed_forward/intermediate_dense/kernel"), F,
                         ffn_sh_only ? nullptr : m->ffn, sh ? m->ffn16 : nullptr, F, 0, m->P(b + "/feed_forward/intermediate_dense/bias"),
                         nullptr, (int)BT, F, H, 1, act, p_f1 ? &m->ffn_in48 : nullptr, p_f2 ? &m->ffn48 : nullptr, keep))
            return e;
        // output dense + residual; StochasticDepth at inference is a plain add (tensorflow_addons.py:386-390)
        float* dst = prenorm ? m->hs[i + 1] : m->t3;
        if (int e = gemm(m->ffn, sh ? m->ffn16 : nullptr, F, 0, m->P(b + "/feed_forward/output_dense/kernel"), H, dst, nullptr, H, 0,
                         m->P(b + "/feed_forward/output_dense/bias"), ffn_res, (int)BT, H, F, 1, 0, p_f2 ? &m->ffn48 : nullptr))
            return e;
        if (!prenorm)       // (its planes are the next layer's attention input)
            if (int e = launch_layer_norm_x(pf, m->t3, m->hs[i + 1], m->P(b + "/final_layer_norm/gamma"),
                                            m->P(b + "/final_layer_norm/beta"), BT, H, eps, 0, sh ? m->hs16[i + 1] : nullptr, s,
                                            (p_qkv && i + 1 < c.num_layers) ? &po_attn : nullptr))
                return e;
    }
    const uint16_t* head_in16 = sh && !prenorm ? m->hs16[c.num_layers] : nullptr;
    if (prenorm) {
        if (int e = launch_layer_norm_x(pf, m->hs[c.num_layers], m->enc_out, m->P("encoder/layer_norm/gamma"),
                                        m->P("encoder/layer_norm/beta"), BT, H, eps, 0, sh ? m->enc16 : nullptr, s))
            return e;
        head_in16 = sh ? m->enc16 : nullptr;
    }
    // ---- head (modeling.py:253-254) ----
    if (c.with_lm_head) {
        if (int e = gemm(m->enc_out, head_in16, H, 0, m->P("lm_head/kernel"), c.vocab_size, out, nullptr, c.vocab_size, 0,
                         m->P("lm_head/bias"), nullptr, (int)BT, c.vocab_size, H, 1, 0))
            return e;
    } else {
        W2V2_HIP_CHECK(hipMemcpyAsync(out, m->enc_out, (size_t)BT * H * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return W2V2_OK;
}

int w2v2_ctc_loss(const float* logits, int32_t B, int32_t T, int32_t V, const int32_t* labels, int32_t U,
                  const int32_t* label_length, const int32_t* logit_length, int32_t blank, float* nll,
                  float* grad, void* stream) {
    return launch_ctc(nullptr, logits, B, T, V, labels, U, label_length, logit_length, blank, nll, grad,
                      reinterpret_cast<hipStream_t>(stream));
}

int w2v2_ctc_loss_fused(const float* logits, int32_t B, int32_t T, int32_t V, const int32_t* labels, int32_t U, int32_t logit_length_all,
                        int32_t blank, float division_factor, float* nll, float* grad, float* loss_sum, void* stream) {
    W2V2_REQUIRE(logit_length_all > 0, "ctc_loss_fused: logit_length_all must be positive");
    W2V2_REQUIRE(division_factor > 0.f, "ctc_loss_fused: division_factor must be positive");
    return launch_ctc_x(w2v2::tl_step_prof, logits, B, T, V, labels, U, nullptr, nullptr, logit_length_all, blank, division_factor, nll, grad, loss_sum,
                        reinterpret_cast<hipStream_t>(stream));
}

int w2v2_activation_info(const w2v2_model* m, const char* name, int64_t shape[3]) {
    W2V2_REQUIRE(m && name && shape, "activation_info: null argument");
    auto it = m->acts.find(name);
    if (it == m->acts.end()) {
        set_error("activation `%s` does not exist (run a forward first)", name);
        return W2V2_ENOTFOUND;
    }
    for (int i = 0; i < 3; ++i) shape[i] = it->second.shape[i];
    return W2V2_OK;
}

int w2v2_copy_activation(w2v2_model* m, const char* name, float* host_dst, int64_t numel, void* stream) {
    W2V2_REQUIRE(m && name && host_dst, "copy_activation: null argument");
    auto it = m->acts.find(name);
    if (it == m->acts.end()) {
        set_error("activation `%s` does not exist (run a forward first)", name);
        return W2V2_ENOTFOUND;
    }
    const Act& a = it->second;
    W2V2_REQUIRE(numel == a.shape[0] * a.shape[1] * a.shape[2], "copy_activation: `%s` element count mismatch", name);
    for (const std::string& skipped : m->acts_skipped)
        if (skipped == name) {
            set_error("activation `%s` was written only as bf16 by the last forward (precision mode bf16: its one consumer reads the "
                      "shadow); w2v2_set_option(m, W2V2_OPT_KEEP_ACTIVATIONS, 1) keeps the fp32 copy", name);
            return W2V2_ESTATE;
        }
    W2V2_HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    W2V2_HIP_CHECK(hipMemcpy(host_dst, a.ptr, (size_t)numel * sizeof(float), hipMemcpyDeviceToHost));
    return W2V2_OK;
}

int w2v2_profile_enable(w2v2_model* m, int enable) {
    W2V2_REQUIRE(m, "profile_enable: null model");
    profiler_enable(m->prof, enable != 0);
    return W2V2_OK;
}
int w2v2_profile_families(w2v2_model* m, uint32_t family_mask) {
    W2V2_REQUIRE(m, "profile_families: null model");
    profiler_set_mask(m->prof, family_mask);
    return W2V2_OK;
}
int w2v2_profile_sampling(w2v2_model* m, int32_t stride) {
    W2V2_REQUIRE(m && stride >= 1, "profile_sampling: bad argument");
    profiler_set_stride(m->prof, stride);
    return W2V2_OK;
}
int w2v2_profile_seen(w2v2_model* m, int index, int64_t* launches) {
    W2V2_REQUIRE(m && index >= 0 && index < FAM_COUNT && launches, "profile_seen: bad argument");
    *launches = profiler_seen(m->prof, index);
    return W2V2_OK;
}
int w2v2_profile_num_families(void) { return FAM_COUNT; }
int w2v2_profile_read(w2v2_model* m, int index, const char** name, int64_t* launches, double* total_ms,
                      double* flops, double* bytes) {
    W2V2_REQUIRE(m && index >= 0 && index < FAM_COUNT && launches && total_ms && flops && bytes, "profile_read: bad argument");
    if (name) *name = family_name(index);
    return profiler_read(m->prof, index, launches, total_ms, flops, bytes);
}
int w2v2_profile_kernel_launches(w2v2_model* m, int index, int64_t* launches) {
    W2V2_REQUIRE(m && launches, "profile_kernel_launches: null argument");
    W2V2_REQUIRE(index >= 0 && index < FAM_COUNT, "profile_kernel_launches: bad family index %d", index);
    *launches = profiler_kernel_launches(m->prof, index);
    return W2V2_OK;
}
int w2v2_profile_reset(w2v2_model* m) {
    W2V2_REQUIRE(m, "profile_reset: null model");
    profiler_reset(m->prof);
    return W2V2_OK;
}

// ---- single operators ---------------------------------------------------------
int w2v2_op_gemm(const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb, float* C,
                 int64_t ldc, int64_t strideC, const float* bias, const float* residual, int32_t M,
                 int32_t N, int32_t K, int32_t nbatch, int32_t act, void* stream) {
    return launch_gemm(nullptr, A, lda, strideA, B, ldb, C, ldc, strideC, bias, residual, M, N, K, nbatch, act,
                       reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_set_precision(int32_t mode) {
    W2V2_REQUIRE(mode >= W2V2_PRECISION_FP32 && mode <= W2V2_PRECISION_F16X2, "op_set_precision: unknown mode %d", mode);
    gemm_set_precision(mode);
    return W2V2_OK;
}
int w2v2_op_gemm_bf16(const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb, float* C,
                      int64_t ldc, int64_t strideC, const float* bias, const float* residual, int32_t M,
                      int32_t N, int32_t K, int32_t nbatch, int32_t act, void* stream) {
    return launch_gemm_bf16(nullptr, A, lda, strideA, B, ldb, 0, C, ldc, strideC, bias, residual, M, N, K, nbatch, act,
                            reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_gemm_bf16_shadows(const uint16_t* A16, int64_t lda, int64_t strideA, const uint16_t* B16, float* C, uint16_t* C16, int64_t ldc,
                              int64_t strideC, const float* bias, const float* residual, int32_t M, int32_t N, int32_t K, int32_t nbatch,
                              int32_t act, int32_t variant, void* stream) {
    W2V2_REQUIRE(A16 && B16 && (C || C16), "op_gemm_bf16_shadows: null operand");
    W2V2_REQUIRE(variant >= 0 && variant <= 2, "op_gemm_bf16_shadows: variant %d (0 by shape, 1 = 128 x 128 tiles, 2 = 128 x 256 software-pipelined)", variant);
    W2V2_REQUIRE(variant != 2 || gemm_bf16_sw_ok(M, N, K, lda, K, strideA),
                 "op_gemm_bf16_shadows: variant 2 needs N %% 256 == 0, K %% 64 == 0, K >= 192 and 16-byte aligned rows");
    GemmShadows x;
    x.A16 = A16; x.B16 = B16; x.C16 = C16; x.ldb16 = K; x.force_kernel = variant;
    return launch_gemm_bf16_x(nullptr, nullptr, lda, strideA, nullptr, N, 0, C, ldc, strideC, bias, residual, M, N, K, nbatch, act, x,
                              reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_gemm_split(const float* A, int64_t lda, int64_t strideA, const float* B, float* C, int64_t ldc, int64_t strideC,
                       const float* bias, const float* residual, int32_t M, int32_t N, int32_t K, int32_t nbatch, int32_t act,
                       void* stream) {
    W2V2_REQUIRE(A && B && C && N > 0 && K > 0, "op_gemm_split: null operand");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    uint16_t* planes = nullptr;
    W2V2_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&planes), (size_t)3 * K * N * sizeof(uint16_t)));
    int e = launch_split_weight(B, planes, K, N, s);
    if (!e) e = launch_gemm_split(nullptr, A, lda, strideA, planes, C, ldc, strideC, bias, residual, M, N, K, nbatch, act, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(planes);
    return e;
}
int w2v2_op_check_select_forms(uint64_t* mismatches, void* stream) {
    return launch_check_select_forms(reinterpret_cast<unsigned long long*>(mismatches), reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_split_planes(const float* x, uint16_t* planes, int64_t plane_stride, int64_t n, int32_t fmt, int32_t* range_flag, void* stream) {
    W2V2_REQUIRE(fmt == PF_BF16X3 || fmt == PF_F16X2, "op_split_planes: unknown plane format %d", fmt);
    return launch_split_planes(x, planes, plane_stride, n, fmt, range_flag, reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_split_weight(const float* B, uint16_t* images, float* scale_ws, int32_t K, int32_t N, int32_t fmt, void* stream) {
    W2V2_REQUIRE(fmt == PF_BF16X3 || fmt == PF_F16X2, "op_split_weight: unknown plane format %d", fmt);
    return launch_split_weight_sw(B, images, K, N, fmt, scale_ws, reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_gemm_split_planes(int32_t fmt, const uint16_t* A16, int64_t planeA, int64_t lda, int64_t strideA, const uint16_t* Bimg,
                              const float* out_scale, float* C, uint16_t* C16, int64_t planeC, int64_t ldc, int64_t strideC,
                              const float* bias, const float* residual, int32_t M, int32_t N, int32_t K, int32_t nbatch, int32_t act,
                              int32_t* range_flag, void* stream) {
    return launch_gemm_split_sw(nullptr, fmt, A16, planeA, lda, strideA, Bimg, out_scale, C, C16, planeC, ldc, strideC, bias, residual, M, N, K,
                                nbatch, act, range_flag, reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_gemm_bf16_at(const float* At, int64_t lda, int64_t strideA, const float* B, int64_t ldb, int64_t strideB,
                         float* C, int64_t ldc, int64_t strideC, int32_t M, int32_t N, int32_t K, int32_t nbatch, void* stream) {
    GemmShadows x;
    x.transA = true;
    return launch_gemm_bf16_x(nullptr, At, lda, strideA, B, ldb, strideB, C, ldc, strideC, nullptr, nullptr, M, N, K, nbatch, 0, x,
                              reinterpret_cast<hipStream_t>(stream));
}
// slicing-by-8 CRC-32C (reflected polynomial 0x82F63B78): ~1 GB/s on the host, a base checkpoint is 0.38 GB
uint32_t w2v2_crc32c_extend(uint32_t crc, const void* data, uint64_t n) {
    static uint32_t T[8][256];
    static std::atomic<bool> ready{false};
    static std::mutex mu;
    if (!ready) {
        std::lock_guard<std::mutex> lk(mu);
        if (!ready) {
            for (uint32_t i = 0; i < 256; ++i) {
                uint32_t c = i;
                for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
                T[0][i] = c;
            }
            for (uint32_t i = 0; i < 256; ++i)
                for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFFu];
            ready = true;
        }
    }
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = crc ^ 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = T[7][lo & 0xFFu] ^ T[6][(lo >> 8) & 0xFFu] ^ T[5][(lo >> 16) & 0xFFu] ^ T[4][lo >> 24] ^ T[3][hi & 0xFFu] ^
            T[2][(hi >> 8) & 0xFFu] ^ T[1][(hi >> 16) & 0xFFu] ^ T[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = T[0][(c ^ *p++) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
int w2v2_op_weight_grad_bf16(const uint16_t* x16, const uint16_t* dy16, float* slabs, int64_t rows, int32_t Kin, int32_t Nout,
                             int32_t rows_per_slab, int32_t nslabs, int32_t variant, void* stream) {
    W2V2_REQUIRE(x16 && dy16 && slabs && rows > 0 && rows_per_slab > 0 && nslabs > 0 && Kin % 128 == 0 && Nout % 128 == 0 &&
                     rows_per_slab % 64 == 0 && variant >= 0 && variant <= 3, "op_weight_grad_bf16: bad argument");
    // variants 2 and 3 also take uneven slabs: with u = ceil(rows / 64) K tiles in all, u - nslabs (rows_per_slab / 64) = e, 0 <= e < nslabs,
    // gives the first e slabs one K tile more (GemmShadows::kextra -- how the training step cuts B T rows into any number of slabs).
    // variant 3 additionally promises that row `rows` of dy16 exists and is all-zero, which lets the last K tile be short.
    const bool sw = variant == 2 || variant == 3;
    const int64_t units = (rows + 63) / 64, over_units = units - (int64_t)(rows_per_slab / 64) * nslabs;
    const bool shape_ok = sw && over_units >= 0 && over_units < nslabs && (rows % 64 == 0 || variant == 3) && (over_units == 0 || nslabs > 1);
    W2V2_REQUIRE(!sw || (shape_ok && gemm_bf16_swtr_ok(Kin, Nout, rows_per_slab, Kin, Nout, (int64_t)rows_per_slab * Kin, (int64_t)rows_per_slab * Nout)),
                 "op_weight_grad_bf16: variants 2 / 3 need whole 64-row K tiles (variant 3: a zero row behind dy16 instead), at most one extra K tile "
                 "per slab, Nout %% 256 == 0 and at least 192 rows per slab");
    GemmShadows x;
    x.transA = true; x.A16 = x16; x.B16p = dy16; x.force_kernel = sw ? 2 : variant;
    if (sw) {
        x.kextra = (int)over_units;
        x.validK = rows % 64 == 0 ? 0 : rows;
        x.b_zero_row = variant == 3;
    } else {
        x.validK = rows == (int64_t)rows_per_slab * nslabs ? 0 : rows;
    }
    return launch_gemm_bf16_x(nullptr, nullptr, Kin, (int64_t)rows_per_slab * Kin, nullptr, Nout, (int64_t)rows_per_slab * Nout, slabs, Nout,
                              (int64_t)Kin * Nout, nullptr, nullptr, Kin, Nout, rows_per_slab, nslabs, 0, x, reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_layer_norm(const float* x, float* y, const float* gamma, const float* beta, int64_t rows,
                       int32_t C, float eps, int32_t act, void* stream) {
    return launch_layer_norm(nullptr, x, y, gamma, beta, rows, C, eps, act, reinterpret_cast<hipStream_t>(stream));
}
int64_t w2v2_conv0_ws_floats(int32_t B, int64_t L, int32_t K, int32_t stride, int32_t C) {
    return conv0_ws_floats(B, L, K, stride, C);
}
int w2v2_op_conv0(const float* wave, const float* kernel, const float* bias, const float* gamma,
                  const float* beta, float* out, float* ws, int32_t B, int64_t L, int32_t K, int32_t stride,
                  int32_t C, float eps, int32_t norm_mode, int32_t act, void* stream) {
    return launch_conv0(nullptr, wave, kernel, bias, gamma, beta, out, ws, B, L, K, stride, C, eps, norm_mode, act,
                        reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_weight_norm_regroup(const float* wv, const float* wg, float* out, int32_t K, int32_t cg,
                                int32_t H, int32_t groups, void* stream) {
    return launch_weight_norm_regroup(nullptr, wv, wg, out, K, cg, H, groups, reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_pos_conv(const float* x, const float* wg, const float* bias, const int32_t* frame_len, float* y,
                     int32_t B, int32_t T, int32_t H, int32_t K, int32_t groups, int32_t act, void* stream) {
    return launch_pos_conv(nullptr, x, wg, bias, frame_len, y, B, T, H, K, groups, act, reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_attention(const float* qkv, const int32_t* frame_len, float* ctx, int32_t B, int32_t T,
                      int32_t H, int32_t num_heads, void* stream) {
    return launch_attention(nullptr, qkv, frame_len, ctx, B, T, H, num_heads, reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_frame_lengths(const int32_t* mask, int32_t* frame_len, int32_t B, int64_t L,
                          const int32_t* kernal_sizes, const int32_t* strides, int32_t num_layers, void* stream) {
    return launch_frame_lengths(nullptr, mask, frame_len, B, L, kernal_sizes, strides, num_layers,
                                reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
