// Training step of the CTC fine-tuning path: training-mode forward, backward, Adam.
//
// What the reference does inside Keras' train_step (src/main.py:136-259):
//   forward with training=True  -- dropout at every Dropout layer, spec-augment on the projected
//     features (modeling.py:193-199), StochasticDepth on the FFN branch (encoder.py:130);
//   CTC loss / global batch (losses.py:45, main.py:198-200);
//   gradients of every trainable variable -- stage 2 freezes the 7 conv layers (main.py:234-237),
//     stage 1 trains lm_head only (main.py:210);
//   Adam (Keras defaults) and, under MirroredStrategy, a SUM all-reduce of the gradients.
// Here: w2v2_train_forward -> (caller: w2v2_ctc_loss gives d nll / d logits) -> w2v2_train_backward
// fills one flat gradient buffer (the all-reduce payload) -> w2v2_adam_step.
//
// Heavy contractions reuse the forward's fp32 MFMA GEMM:
//   dX = dY W^T   uses transposed copies of the kernels kept next to the variables;
//   dW = X^T dY   transposes the activation once and runs a split-K batched GEMM into slabs that a
//                 reduction kernel sums (deterministic: no atomics anywhere in the backward).
// The conv feature extractor has no backward: the reference never trains it in this path.
#include <string.h>

#include "model.h"
#include "train.h"

using namespace w2v2;

struct LayerSave {
    float *qkv, *ctx, *lse, *t1, *t2, *u, *gd, *t3;
    float* a;                           // prenorm only: LN(x), the attention input
    float *WqkvT, *WoT, *W1T, *W2T;     // transposed kernels for the data-gradient GEMMs
    float keep;                         // stochastic-depth draw of the last forward
    // precision mode 1: bf16 shadows of the activations the weight-gradient GEMMs contract with (dW = X^T dY reads X as a
    // transposed A).  Written by the forward's producers like the shared shadows, but kept per layer until the backward.
    uint16_t *a16 = nullptr, *ctx16 = nullptr, *t2_16 = nullptr, *gd16 = nullptr, *qkv16 = nullptr;
    uint32_t* keep_bits = nullptr;       // attention-probability dropout decisions of the forward (bf16 attention kernels)
};

struct TrainState {
    int B = 0, T = 0;
    int64_t L = 0;
    int lean = 0;                       // LEAN_* set the shape-dependent buffers were built with (ensure_train_ws)
    int64_t alloc_bytes = 0;            // bytes behind `allocs`
    std::vector<void*> allocs;          // shape-dependent buffers: rebuilt when (B, L) changes
    std::vector<void*> persist;         // shape-independent buffers (gradients, Adam moments, transposed kernels, slab scratch):
                                        // allocated once -- their addresses key the bf16x3 plane cache (w2v2_model::w48)
    std::vector<LayerSave> layers;
    float *hd = nullptr, *hm = nullptr, *pos_c = nullptr, *hdf = nullptr, *hs0 = nullptr;
    float *WpT = nullptr, *WlmT = nullptr, *pos_wg_t = nullptr, *dwg = nullptr;
    float *pos_pack32 = nullptr, *pos_dw_slabs = nullptr, *pos_dc_pad = nullptr;   // scratch of the GEMM-formulated positional-conv kernel gradient
    uint16_t* pos_w16_t = nullptr;           // bf16 (groups, og, K cg) shadow of pos_wg_t (precision mode 1); follows transposes_fresh
    bool pos_w16_t_fresh = false;
    uint8_t* spec_mask = nullptr;       // (B*T) device copy, or null when not applied
    bool have_spec = false, have_mask = false;
    uint64_t trainable_sig = 0;           // signature of `trainable` at the last backward (0: none yet -> the buffer is cleared)
    bool grads_need_clear = false;
    float p = 0.f;
    uint64_t seed = 0;
    // gradient / optimizer state: flat, inventory order
    float *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    AdamChunk* adam_chunks = nullptr;        // device table for the single-launch Adam; rebuilt when `trainable` changes
    int adam_nchunks = 0;
    bool adam_table_fresh = false;
    std::vector<int64_t> goff;
    int64_t gtotal = 0;
    std::vector<char> trainable;
    bool transposes_fresh = false;
    bool transposes_full = false;            // the per-layer fp32 transposed kernels are current (not needed in bf16 mode)
    // scratch
    float *gh[4] = {nullptr, nullptr, nullptr, nullptr}, *gf = nullptr, *g3h = nullptr, *at = nullptr,
          *slabs = nullptr, *red_ws = nullptr, *dvec = nullptr, *dummy = nullptr, *dwqkv = nullptr, *dwv_scratch = nullptr,
          *attn_colpart = nullptr;     // per-block column sums of dqkv from the bf16 attention backward (-> q|k|v bias gradient)
    int64_t slab_floats = 0;
    // bf16 shadows of the gradient tensors that are the A operand of a data-gradient GEMM (precision mode 1): written by the
    // producing kernel (LayerNorm / dropout / attention backward), consumed by the GEMM enqueued right behind it
    uint16_t *dy16_h = nullptr, *dy16_f = nullptr, *dy16_3h = nullptr, *dy16_ctx = nullptr;
    float* cs_ws = nullptr;           // (slabs + 1, widest N): per-slab column sums of dY from the weight-gradient GEMM
    int64_t cs_floats = 0;
    bool forward_done = false;
    bool ffn16_only = false;                      // the last forward wrote dropout(GELU(u)) only as bf16 (no fp32 l.gd)
    bool ctx16_only = false;                      // ... and the attention output only as bf16 (no fp32 l.ctx: the backward reads O and dO as bf16)
    bool u16_only = false;                        // ... and the FFN pre-activation u only as bf16 (in the first half of l.u's storage)
    bool ln16_only = false;                       // ... and (prenorm) the two in-layer LayerNorm outputs a, t2 only as bf16
    bool x16_valid = false, x16_attn = false;     // the last forward wrote the per-layer bf16 shadows (/ ctx16 from the bf16 attention)
    // gradient buckets for overlapping the data-parallel all-reduce with the backward: completion order
    //   0 = lm_head, 1 .. N = encoder layers N-1 .. 0, N+1 = everything in front of layer 0 in the flat buffer
    std::vector<hipEvent_t> bucket_ev;
    // Weight-gradient side stream (round 5, W2V2_OPT_WGRAD_STREAM, default OFF).  dW = X^T dY is off the backward's critical path --
    // only the optimizer (and the bucket's all-reduce) reads it -- so the encoder layers' four weight-gradient GEMMs (+ their slab
    // folds) can be enqueued on a second, lower-priority stream, to fill what the critical path leaves idle: the underfilled last
    // round of every N = 768 data-gradient GEMM (576 tiles on 512 block slots), the matrix pipe under the VALU-bound attention
    // backward, the HBM-bound element-wise passes.  MEASURED (profiles/r05_ab_wgrad_stream.txt): the kernels do overlap -- every
    // family's event brackets stretch -- and the step time does not move (33.38 vs 33.39 ms base, 97.7 vs 98.0 large-robust): the
    // chip is a shared-throughput machine here (clock 1.98 GHz under the bf16 GEMMs: power; HBM under the element-wise passes), not a
    // slot-limited one.  Kept as an option because the all-reduce overlap of an 8-GPU job may still want the earlier buckets; off by
    // default.  Ordering is by events only: `wg_fork` (main -> side: the dY the GEMM reads is complete), `wg_join[site]` (side -> main,
    // waited for right before the main stream next overwrites that site's dY), and the bucket events are recorded on the side stream
    // after it has also waited for the main stream's position.  Same kernels, same operands: results are bit-identical to one stream.
    hipStream_t wg_stream = nullptr;
    hipEvent_t wg_fork = nullptr, wg_main = nullptr, wg_join[4] = {nullptr, nullptr, nullptr, nullptr};
    float* red_ws_side = nullptr;     // the side stream's own reduction scratch (red_ws belongs to the main stream's kernels)
    // Deferred folds (train.h: FoldBatch): the partial rows of a layer's gradients live until the layer's one fold launch, so each
    // deferred producer has its own scratch -- split-K slabs of the four weight gradients (0 = FFN down, 1 = FFN up, 2 = out-projection,
    // 3 = q|k|v; 33 x the kernel's size each: ~0.9 GB for base, 1.7 GB for large) and the per-block column sums of LayerNorm 2 / the FFN
    // dropout backward / LayerNorm 1 (site_ws).  The attention backward's column partials already have theirs (attn_colpart).
    float* site_slabs[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t site_slab_floats[4] = {0, 0, 0, 0};
    float* site_ws[3] = {nullptr, nullptr, nullptr};
    int64_t red_ws_floats = 0;        // size of red_ws (and of each of red_ws_side / site_ws[k] once they exist)
};

// Slabs a weight gradient of `elems` = Kin x Nout elements can be cut into: weight_grad's rule is tiles x S <= 512 blocks for the bf16
// kernels (tiles of at most 128 x 256) and <= 2048 for the fp32 one (128 x 128), S <= 32 -- plus one slab of fold scratch.  (Round 5
// allocated 33 slabs for every site whatever its shape: 0.93 GB for base, 1.66 GB for large, where this rule gives 0.50 / 0.59 GB.)
static int64_t site_slab_count(int64_t elems) {
    const int64_t s = ((int64_t)2048 * 16384 + elems - 1) / elems;
    return (s < 32 ? s : 32) + 1;
}

// where a weight gradient puts its split-K slabs and who folds them
struct WgSite {
    float* slabs = nullptr;           // own slab scratch (TrainState::site_slabs[k]); null: the shared TrainState::slabs
    int64_t slab_floats = 0;          // its size
    FoldBatch* defer = nullptr;       // the slab fold joins this batch instead of running behind the GEMM
    float* unpack3[3] = {nullptr, nullptr, nullptr};      // q|k|v: the fold writes the three (H, H) kernels straight from the packed (H, 3H) slabs
    int unpackH = 0;
};

static int t_alloc(TrainState* t, float** out, int64_t floats) {
    void* p = nullptr;
    W2V2_HIP_CHECK(hipMalloc(&p, (size_t)(floats > 0 ? floats : 1) * sizeof(float)));
    t->allocs.push_back(p);
    t->alloc_bytes += (floats > 0 ? floats : 1) * (int64_t)sizeof(float);
    *out = reinterpret_cast<float*>(p);
    return W2V2_OK;
}

static int p_alloc(TrainState* t, float** out, int64_t floats) {
    void* p = nullptr;
    W2V2_HIP_CHECK(hipMalloc(&p, (size_t)(floats > 0 ? floats : 1) * sizeof(float)));
    t->persist.push_back(p);
    *out = reinterpret_cast<float*>(p);
    return W2V2_OK;
}

static void t_free(TrainState* t) {
    for (void* p : t->allocs) (void)hipFree(p);
    t->allocs.clear();
    t->alloc_bytes = 0;
}

void w2v2_train_destroy(w2v2_model* m) {
    if (!m || !m->train) return;
    t_free(m->train);
    for (void* p : m->train->persist) (void)hipFree(p);
    m->train->persist.clear();
    if (m->train->adam_chunks) (void)hipFree(m->train->adam_chunks);
    if (m->train->pos_w16_t) (void)hipFree(m->train->pos_w16_t);
    for (hipEvent_t ev : m->train->bucket_ev) (void)hipEventDestroy(ev);
    if (m->train->wg_stream) (void)hipStreamDestroy(m->train->wg_stream);
    for (hipEvent_t ev : {m->train->wg_fork, m->train->wg_main, m->train->wg_join[0], m->train->wg_join[1], m->train->wg_join[2], m->train->wg_join[3]})
        if (ev) (void)hipEventDestroy(ev);
    delete m->train;
    m->train = nullptr;
}

// a variable changed outside the optimizer (w2v2_set_param): the transposed kernel copies are stale
void w2v2_train_invalidate(w2v2_model* m) {
    if (m && m->train) m->train->transposes_fresh = false;
}

static TrainState* get_state(w2v2_model* m) {
    if (!m->train) {
        TrainState* t = new TrainState();
        t->trainable.assign(m->params.size(), 1);
        // flat gradient layout = inventory order
        t->goff.resize(m->params.size());
        int64_t off = 0;
        for (size_t i = 0; i < m->params.size(); ++i) {
            t->goff[i] = off;
            off += (m->params[i].numel + 3) & ~(int64_t)3;     // 16-byte aligned slots
        }
        t->gtotal = off;
        m->train = t;
    }
    return m->train;
}

// Shape-independent state, once per model: gradients, Adam moments, transposed kernel copies, slab scratch.  The transposed
// copies must keep their addresses across shape changes: the bf16x3 plane cache is keyed by them, and a freed-and-recycled
// address would serve stale planes.
static int ensure_persistent(w2v2_model* m) {
    TrainState* t = get_state(m);
    const w2v2_config& c = m->cfg;
    const int64_t H = c.hidden_size, F = c.intermediate_size;
    const int64_t C = c.filter_sizes[c.num_conv_layers - 1];
    const int64_t K = c.num_conv_pos_embeddings, cg = H / c.num_conv_pos_embedding_groups;
    if (!t->grads) {
        if (int e = p_alloc(t, &t->grads, t->gtotal)) return e;
        if (int e = p_alloc(t, &t->adam_m, t->gtotal)) return e;
        if (int e = p_alloc(t, &t->adam_v, t->gtotal)) return e;
        W2V2_HIP_CHECK(hipMemset(t->grads, 0, (size_t)t->gtotal * 4));
        W2V2_HIP_CHECK(hipMemset(t->adam_m, 0, (size_t)t->gtotal * 4));
        W2V2_HIP_CHECK(hipMemset(t->adam_v, 0, (size_t)t->gtotal * 4));
        if (int e = p_alloc(t, &t->WpT, H * C)) return e;
        if (int e = p_alloc(t, &t->WlmT, H * (int64_t)c.vocab_size)) return e;
        if (int e = p_alloc(t, &t->pos_wg_t, K * cg * H)) return e;
        if (int e = p_alloc(t, &t->dwg, K * cg * H)) return e;
        if (int e = p_alloc(t, &t->dwv_scratch, K * cg * H)) return e;
        t->layers.resize(c.num_layers);
        for (auto& l : t->layers) {
            if (int e = p_alloc(t, &l.WqkvT, 3 * H * H)) return e;
            if (int e = p_alloc(t, &l.WoT, H * H)) return e;
            if (int e = p_alloc(t, &l.W1T, F * H)) return e;
            if (int e = p_alloc(t, &l.W2T, H * F)) return e;
            l.keep = 1.f;
        }
        t->slab_floats = 33 * (F * H > 3 * H * H ? F * H : 3 * H * H);      // 32 split-K slabs + the reduction scratch
        if (int e = p_alloc(t, &t->slabs, t->slab_floats)) return e;
        // (the per-site slab scratches of the deferred folds are allocated by the first backward that defers: ensure_site_scratch)
        t->cs_floats = 34 * (F > 3 * H ? F : 3 * H);
        if (int e = p_alloc(t, &t->cs_ws, t->cs_floats)) return e;
        if (int e = p_alloc(t, &t->dwqkv, 3 * H * H + 3 * H)) return e;
        if (int e = p_alloc(t, &t->dummy, 2 * (H + C + F))) return e;       // sink for gradients of frozen LN params
        t->transposes_fresh = false;
    }
    return W2V2_OK;
}

// Scratch that only some backward configurations use, allocated the first time one of them runs (it used to be allocated
// for every model -- fp32 training, W2V2_OPT_DEFER_FOLDS = 0 and the side-stream mode included): the side stream's reduction scratch,
// the deferred folds' per-site partial rows (shape-dependent: freed with the workspace) and per-site split-K slabs (persistent).
static int ensure_site_scratch(w2v2_model* m, bool side, bool defer_cols, bool defer_slabs) {
    TrainState* t = m->train;
    if (side && !t->red_ws_side)
        if (int e = t_alloc(t, &t->red_ws_side, t->red_ws_floats)) return e;
    if (defer_cols && !t->site_ws[0])
        for (int k = 0; k < 3; ++k)
            if (int e = t_alloc(t, &t->site_ws[k], t->red_ws_floats)) return e;
    if (defer_slabs && !t->site_slabs[0]) {
        const int64_t H = m->cfg.hidden_size, F = m->cfg.intermediate_size;
        const int64_t site_elems[4] = {F * H, H * F, H * H, 3 * H * H};
        for (int k = 0; k < 4; ++k) {
            t->site_slab_floats[k] = site_slab_count(site_elems[k]) * site_elems[k];
            if (int e = p_alloc(t, &t->site_slabs[k], t->site_slab_floats[k])) return e;
        }
    }
    return W2V2_OK;
}

// `lean`: which per-layer fp32 activations the coming forward will NOT write (precision mode 1 on the shadow paths keeps q|k|v, the
// attention output and the FFN hidden activation only as bf16, and u as bf16 in half of its buffer): they are not allocated -- 0.75 GB
// per layer at 32 x 246000, 9 GB for base and 24 GB for large at 16 x 480000 (rounds 2-4 allocated them regardless).  A forward that
// needs a different set (the precision or the shadow option changed on the same shapes) rebuilds the workspace.
// LEAN_LN (prenorm only): the outputs of the two LayerNorms inside a layer feed nothing but GEMMs -- q|k|v / the FFN up-projection and their
// weight gradients, which all stream the bf16 shadow -- so their fp32 copies (a, t2) are neither written nor allocated.  (Postnorm: t2 is
// also the FFN's residual and stays fp32.)
enum : int { LEAN_QKV = 1, LEAN_CTX = 2, LEAN_GD = 4, LEAN_U_HALF = 8, LEAN_LN = 16 };
static int ensure_train_ws(w2v2_model* m, int B, int64_t L, int T, int lean) {
    TrainState* t = get_state(m);
    if (t->B == B && t->L == L && t->B > 0 && t->lean == lean) return W2V2_OK;
    if (int e = ensure_persistent(m)) return e;
    const w2v2_config& c = m->cfg;
    const int64_t H = c.hidden_size, F = c.intermediate_size, BT = (int64_t)B * T;
    const int64_t C = c.filter_sizes[c.num_conv_layers - 1];
    // ---- shape-dependent activations and scratch
    t_free(t);
    t->B = 0;
    t->forward_done = false;
    t->pos_pack32 = t->pos_dw_slabs = t->pos_dc_pad = nullptr;      // lazily re-allocated at the new shape
    if (int e = t_alloc(t, &t->hd, BT * H)) return e;
    if (int e = t_alloc(t, &t->hm, BT * H)) return e;
    if (int e = t_alloc(t, &t->pos_c, BT * H)) return e;
    if (int e = t_alloc(t, &t->hdf, BT * H)) return e;
    t->hs0 = nullptr;
    if (c.attention_norm_type == 1)
        if (int e = t_alloc(t, &t->hs0, BT * H)) return e;
    float* sm = nullptr;
    if (int e = t_alloc(t, &sm, (BT + 15) / 4 + 4)) return e;
    t->spec_mask = reinterpret_cast<uint8_t*>(sm);
    for (auto& l : t->layers) {
        l.qkv = l.ctx = l.gd = nullptr;
        if (!(lean & LEAN_QKV))
            if (int e = t_alloc(t, &l.qkv, BT * 3 * H)) return e;
        if (!(lean & LEAN_CTX))
            if (int e = t_alloc(t, &l.ctx, BT * H)) return e;
        if (int e = t_alloc(t, &l.lse, (int64_t)B * c.num_heads * T)) return e;
        if (int e = t_alloc(t, &l.t1, BT * H)) return e;
        l.t2 = nullptr;
        if (!(lean & LEAN_LN))
            if (int e = t_alloc(t, &l.t2, BT * H)) return e;
        if (int e = t_alloc(t, &l.u, (lean & LEAN_U_HALF) ? (BT * F + 1) / 2 + 4 : BT * F)) return e;
        if (!(lean & LEAN_GD))
            if (int e = t_alloc(t, &l.gd, BT * F)) return e;
        if (int e = t_alloc(t, &l.t3, BT * H)) return e;
        l.a = nullptr;
        if (c.attention_norm_type == 1 && !(lean & LEAN_LN))
            if (int e = t_alloc(t, &l.a, BT * H)) return e;
    }
    {
        // per-layer X shadows: (3 H + F) bf16 per row and layer (base, B = 32: 2.7 GB)
        auto up8 = [](int64_t n) { return (n + 7) & ~(int64_t)7; };
        for (auto& l : t->layers) {
            float* raw = nullptr;
            if (int e = t_alloc(t, &raw, (6 * up8(BT * H) + up8(BT * F)) / 2 + 16)) return e;
            l.a16 = reinterpret_cast<uint16_t*>(raw);
            l.ctx16 = l.a16 + up8(BT * H);
            l.t2_16 = l.ctx16 + up8(BT * H);
            l.gd16 = l.t2_16 + up8(BT * H);
            l.qkv16 = l.gd16 + up8(BT * F);          // q | k | v as the bf16 attention kernels read it
            float* kb = nullptr;
            if (attention_bf16_supported((int)(H / c.num_heads))) {
                if (int e = t_alloc(t, &kb, attention_keep_bits_words(B, T, c.num_heads))) return e;
            }
            l.keep_bits = reinterpret_cast<uint32_t*>(kb);
        }
    }
    for (int i = 0; i < 4; ++i)
        if (int e = t_alloc(t, &t->gh[i], BT * H)) return e;
    if (int e = t_alloc(t, &t->gf, BT * F)) return e;
    if (int e = t_alloc(t, &t->g3h, BT * 3 * H)) return e;
    const int64_t widest = F > 3 * H ? F : 3 * H;
    if (int e = t_alloc(t, &t->at, widest * BT)) return e;
    int64_t rw = dropout_bwd_colsum_ws_floats(BT, (int)widest);        // (>= colsum_ws_floats of the same shape)
    const int64_t lw = ln_bwd_ws_floats(BT, (int)(H > C ? H : C));
    if (lw > rw) rw = lw;
    if (int e = t_alloc(t, &t->red_ws, rw + 16)) return e;
    t->red_ws_side = nullptr;                    // (the side stream's scratch and the deferred folds' per-site scratch: allocated by the
    t->site_ws[0] = t->site_ws[1] = t->site_ws[2] = nullptr;      //  first backward that uses them -- ensure_site_scratch)
    t->red_ws_floats = rw + 16;
    if (int e = t_alloc(t, &t->dvec, (int64_t)B * c.num_heads * T)) return e;
    t->attn_colpart = nullptr;
    if (attention_bf16_supported((int)(H / c.num_heads)))
        if (int e = t_alloc(t, &t->attn_colpart, (int64_t)attention_colpart_rows(B, T) * 3 * H)) return e;
    {
        // (BT + 1, H) + (BT + 1, F) + (BT + 1, 3H) + (BT, H) bf16, each 16-byte aligned.  Row BT of the first three is never written
        // and stays zero: the weight-gradient GEMMs read a short last K tile's missing rows from there (weight_grad: dy16_zero_row).
        float* raw = nullptr;
        auto up8 = [](int64_t n) { return (n + 7) & ~(int64_t)7; };
        const int64_t words = (up8((BT + 1) * H) + up8((BT + 1) * F) + up8((BT + 1) * 3 * H) + up8(BT * H)) / 2 + 16;
        if (int e = t_alloc(t, &raw, words)) return e;
        W2V2_HIP_CHECK(hipMemset(raw, 0, (size_t)words * 4));
        W2V2_HIP_CHECK(hipDeviceSynchronize());      // (a device memset may return before it ran, and the step's stream need not be a blocking one)
        t->dy16_h = reinterpret_cast<uint16_t*>(raw);
        t->dy16_f = t->dy16_h + up8((BT + 1) * H);
        t->dy16_3h = t->dy16_f + up8((BT + 1) * F);
        t->dy16_ctx = t->dy16_3h + up8((BT + 1) * 3 * H);      // dctx as the bf16 attention backward reads it
    }
    t->B = B;
    t->L = L;
    t->T = T;
    t->lean = lean;
    return W2V2_OK;
}

static int refresh_transposes(w2v2_model* m, hipStream_t s) {
    TrainState* t = m->train;
    // Precision mode 1 with shadows takes the plain bf16 copy of W as the (N, K) shadow of W^T in every data-gradient GEMM
    // that has one (all of the transformer's and the projection's), so the fp32 transposed copies of those kernels are
    // never read: only lm_head's (N = 32: no shadow) and the flipped positional kernel are needed.  `transposes_full`
    // remembers whether the fp32 copies are current, for a later switch back to fp32 / bf16x3 on the same model.
    const w2v2_config& c = m->cfg;
    bool need_full = !(m->precision == 1 && w2v2_shadows_enabled(m) && m->w16_valid);
    if (!need_full) {      // every kernel whose fp32 copy is skipped must really have its bf16 stand-in (shapes with N % 64 != 0 do not)
        auto has = [&](const float* w) { return m->w16p.find(w) != m->w16p.end(); };
        need_full = !has(m->P("feature_projection/projection/kernel"));
        for (int i = 0; i < c.num_layers && !need_full; ++i) {
            const std::string b = "encoder/layers/" + std::to_string(i);
            need_full = !(has(m->qkv_w[i]) && has(m->P(b + "/attention/out_proj/kernel")) &&
                          has(m->P(b + "/feed_forward/intermediate_dense/kernel")) && has(m->P(b + "/feed_forward/output_dense/kernel")));
        }
    }
    if (t->transposes_fresh && (t->transposes_full || !need_full)) return W2V2_OK;
    const int H = c.hidden_size, F = c.intermediate_size;
    const int C = c.filter_sizes[c.num_conv_layers - 1];
    if (c.with_lm_head)
        if (int e = launch_transpose(m->P("lm_head/kernel"), t->WlmT, H, c.vocab_size, 1, s)) return e;
    if (need_full) {
        if (int e = launch_transpose(m->P("feature_projection/projection/kernel"), t->WpT, C, H, 1, s)) return e;
        for (int i = 0; i < c.num_layers; ++i) {
            const std::string b = "encoder/layers/" + std::to_string(i);
            LayerSave& l = t->layers[i];
            if (int e = launch_transpose(m->qkv_w[i], l.WqkvT, H, 3 * H, 1, s)) return e;
            if (int e = launch_transpose(m->P(b + "/attention/out_proj/kernel"), l.WoT, H, H, 1, s)) return e;
            if (int e = launch_transpose(m->P(b + "/feed_forward/intermediate_dense/kernel"), l.W1T, H, F, 1, s)) return e;
            if (int e = launch_transpose(m->P(b + "/feed_forward/output_dense/kernel"), l.W2T, F, H, 1, s)) return e;
        }
    }
    const int K = c.num_conv_pos_embeddings, G = c.num_conv_pos_embedding_groups;
    if (int e = launch_pos_conv_flip_regroup(m->pos_wg, t->pos_wg_t, K, H / G, G, s)) return e;
    t->pos_w16_t_fresh = false;
    t->transposes_fresh = true;
    t->transposes_full = need_full;
    return W2V2_OK;
}

static inline uint32_t layer_stream(int layer, int site) { return DS_LAYER_BASE + 4u * (uint32_t)layer + (uint32_t)site; }

static float* grad_of(w2v2_model* m, const std::string& name) {
    auto it = m->index.find(name);
    return it == m->index.end() ? nullptr : m->train->grads + m->train->goff[it->second];
}
static bool is_trainable(w2v2_model* m, const std::string& name) {
    auto it = m->index.find(name);
    return it != m->index.end() && m->train->trainable[it->second];
}

// dW (Kin x Nout) = A^T (Kin x M) dY (M x Nout), db = column sums of dY.  A is (M x Kin) row-major.
// A16 / dY16: bf16 shadows of A and dY (row-major, same shapes) or null: with both, the fast slabs read half the bytes
// dy16_zero_row: the caller keeps row M of dY16 all-zero (TrainState's dY shadows are allocated that way) -- a row count that is not a
// multiple of 64 can then take the 128 x 256 kernel's ragged form instead of the 128 x 128 kernel
static int weight_grad(w2v2_model* m, const float* A, const float* dY, int M, int Kin, int Nout, float* dW,
                       float* db, hipStream_t s, const uint16_t* A16 = nullptr, const uint16_t* dY16 = nullptr, bool dy16_zero_row = false,
                       float* red_ws = nullptr, const WgSite* site = nullptr) {
    TrainState* t = m->train;
    if (!red_ws) red_ws = t->red_ws;       // (the side stream passes its own: TrainState::red_ws_side)
    bool fused_bias = false;
    // slab scratch: the shared one, or this site's own (33 Kin Nout floats: S <= 32 slabs; the deferred fold needs no partial scratch
    // behind them).  The slab-count rule below keeps using TrainState::slab_floats either way, so S does not depend on the site.
    float* const slabs = (site && site->slabs) ? site->slabs : t->slabs;
    FoldBatch* const defer = (site && site->slabs) ? site->defer : nullptr;
    const bool unpack = site && site->unpackH > 0;
    // dW = sum of `nrows` slabs: behind the GEMM, or as a job of the layer's fold launch.  (`unpack`: even a single slab goes through the
    // fold, which scatters the packed matrix into the three kernels)
    auto fold_slabs = [&](int nrows) -> int {
        if (defer && (unpack ? defer->add_tall(slabs, nrows, (int64_t)Kin * Nout, site->unpack3[0], site->unpackH, site->unpack3[1], site->unpack3[2])
                             : defer->add_tall(slabs, nrows, (int64_t)Kin * Nout, dW)))
            return W2V2_OK;
        if (unpack) {           // behind the GEMM, but still one launch for the sum and the scatter into the three kernels
            FoldBatch one;
            W2V2_REQUIRE(one.add_tall(slabs, nrows, (int64_t)Kin * Nout, site->unpack3[0], site->unpackH, site->unpack3[1], site->unpack3[2]),
                         "weight_grad: the unpacking fold does not fit this shape");
            return one.flush(s);
        }
        return launch_colsum(slabs, dW, nrows, Kin * Nout, slabs + (int64_t)nrows * Kin * Nout, 0, s);
    };
    if (dW) {
        // dW (Kin, Nout) = A^T dY over the M = B T rows: few output tiles and a very long K, so the rows are cut into S slabs
        // (one GEMM batch each) that a column sum folds.  precision mode 1 stages A^T from X directly (the bf16 GEMM's B path
        // transposes in registers anyway); fp32 goes through a transposed copy.
        const bool direct = gemm_get_precision() == 1 && Kin % 4 == 0 && Nout % 4 == 0;
        const int kq = direct ? 64 : 32;        // K granularity of the fast kernel that will run
        // The fast kernels need every slab to be a multiple of kq rows.  M need not be (T = 1499 at 480000 samples gives
        // B T = 23984): the first Mq = kq floor(M / kq) rows go through the fast path in S equal slabs, S a divisor of
        // Mq / kq, and the R = M - Mq < kq leftover rows form one more slab on the guarded kernel (0.2 % of the work).
        // (Before this, such an M fell back to ONE guarded GEMM over all rows: 256 tiles, K = 23984.)
        // (tiles as the kernel that will run counts them: with both bf16 shadows, whole rows and Nout % 256 == 0 the weight gradient
        //  takes the 128 x 256 software-pipelined kernel, which has half as many tiles to spread over the 512 block slots)
        const bool wide_tiles = direct && A16 && dY16 && Kin % 128 == 0 && Nout % 256 == 0 && (M % kq == 0 || (dy16_zero_row && M > kq));
        const int64_t tiles = (int64_t)((Kin + 127) / 128) * (wide_tiles ? Nout / 256 : (Nout + 127) / 128);
        // slabs cost a reduction pass each: the bf16 kernels are happiest with ONE block per resident slot (512),
        // the fp32 one wants ~4 to balance its long tiles (183.8 vs 189.4 ms)
        const int dwb = tune_int("W2V2_DW_BLOCKS", 512);      // 512 -> 40.0 ms per step, 1024 -> 40.5, 256 -> 47.0 (base, 32 x 246000)
        const int64_t max_blocks = direct ? dwb : 2048;
        int cap = 32;                                                   // most slabs worth having / that fit the scratch
        while (cap > 1 && (tiles * cap > max_blocks || (int64_t)(cap + 2) * Kin * Nout > t->slab_floats)) --cap;
        // (a site's own scratch is sized for the most slabs the line above can give its shape -- site_slab_count; should that bound ever
        //  be wrong the count shrinks to what fits instead of writing past the end: correct, though no longer the shared scratch's count)
        while (cap > 1 && site && site->slabs && (int64_t)(cap + 1) * Kin * Nout > site->slab_floats) --cap;
        // S must divide the number of kq-row units; if M / kq has no useful divisor (a prime, say), give up to 15 more
        // units to the leftover slab until one appears
        // both bf16 shadows and whole 128 x 128 tiles: the LDS-DMA + transposing-read kernel; it has no fp32 dY in registers,
        // so the bias gradient (a sum of the UNROUNDED dY) takes the column-sum pass below instead of riding along.  In this form
        // the fp32 A / dY are not read at all (callers may pass null when nothing else needs them).
        const bool tr_form = direct && A16 && dY16 && Kin % 128 == 0 && Nout % 128 == 0;
        W2V2_REQUIRE(tr_form || (A && dY), "weight_grad: the fp32 operands are needed here (no bf16 shadows / shapes not whole 128-tiles)");
        // The 128 x 256 transposed kernel: ANY slab count works there (GemmShadows::kextra: the first ceil(M / 64) mod S slabs run one
        // K tile more; a short last K tile reads its missing rows from the zero row behind dY16), so take as many slabs as fit the block slots -- 7 x 72 tiles = 504 blocks for the FFN matrices where
        // the divisor rule below stops at 6 x 72 = 432 (a sixth of the chip idle while the longest CUs run two 64-K-tile blocks).
        if (wide_tiles && tr_form && tune_int("W2V2_DW_UNEVEN", 1) != 0) {
            const int64_t units_all = (M + kq - 1) / kq;                   // (the last unit may be short: GemmShadows::b_zero_row)
            const int S = (int)std::min<int64_t>(cap, units_all / 3);      // (a slab is at least three K tiles: the kernel's pipeline depth)
            if (S >= 1) {
                const int64_t q = units_all / S;
                const int Kp = (int)(q * kq);
                GemmShadows x;
                x.transA = true; x.A16 = A16; x.B16p = dY16; x.kextra = (int)(units_all % S);
                if (M % kq != 0) { x.validK = M; x.b_zero_row = true; }
                if (int e = launch_gemm_bf16_x(m->prof, nullptr, Kin, (int64_t)Kp * Kin, nullptr, Nout, (int64_t)Kp * Nout, (S == 1 && !unpack) ? dW : slabs, Nout,
                                               (int64_t)Kin * Nout, nullptr, nullptr, Kin, Nout, Kp, S, 0, x, s))
                    return e;
                if (S > 1 || unpack)
                    if (int e = fold_slabs(S)) return e;
                if (db) {
                    W2V2_REQUIRE(dY, "weight_grad: the bias gradient needs the fp32 dY");
                    if (int e = launch_colsum(dY, db, M, Nout, red_ws, 0, s)) return e;
                }
                return W2V2_OK;
            }
        }
        // That kernel reads rows past M as zero (GemmShadows::validK), so any M splits into S slabs of ceil(M / 64 / S) K tiles with
        // no leftover pass: the last slab is merely short.
        const bool ragged = !tune_int("W2V2_NO_RAGGED_DW", 0);      // (tuning build: 1 = leftover rows on the tail kernel instead)
        if (tr_form && ragged && M % kq != 0) {
            const int64_t units_all = (M + kq - 1) / kq;
            int S = (int)(units_all < cap ? units_all : cap);
            int64_t per = (units_all + S - 1) / S;
            while (S > 1 && (int64_t)(S - 1) * per >= units_all) { --S; per = (units_all + S - 1) / S; }     // (no empty slab)
            const int Kp = (int)(per * kq);
            W2V2_REQUIRE(S == 1 || (int64_t)(S + 1) * Kin * Nout <= t->slab_floats, "weight_grad: slab scratch too small");
            GemmShadows x;
            x.transA = true; x.A16 = A16; x.B16p = dY16; x.validK = M;
            if (int e = launch_gemm_bf16_x(m->prof, nullptr, Kin, (int64_t)Kp * Kin, nullptr, Nout, (int64_t)Kp * Nout, (S == 1 && !unpack) ? dW : slabs, Nout,
                                           (int64_t)Kin * Nout, nullptr, nullptr, Kin, Nout, Kp, S, 0, x, s))
                return e;
            if (S > 1 || unpack)
                if (int e = fold_slabs(S)) return e;
            if (db) {
                W2V2_REQUIRE(dY, "weight_grad: the bias gradient needs the fp32 dY");
                if (int e = launch_colsum(dY, db, M, Nout, red_ws, 0, s)) return e;
            }
            return W2V2_OK;
        }
        const int64_t units0 = M / kq;
        int64_t units = units0;
        int S = units0 > 0 ? 1 : 0;
        for (int drop = 0; drop < 16 && units0 - drop > 0; ++drop) {
            const int64_t u = units0 - drop;
            int d = 1;
            for (int cand = cap; cand >= 2; --cand)
                if (u % cand == 0) { d = cand; break; }
            if (d > S) { S = d; units = u; }
            if (2 * d >= cap) break;                                    // good enough: stop giving rows away
        }
        const int Mq = (int)(units * kq), R = M - Mq;
        const int nslabs = S + (R ? 1 : 0);
        W2V2_REQUIRE(nslabs == 1 || (int64_t)(nslabs + 1) * Kin * Nout <= t->slab_floats, "weight_grad: slab scratch too small");
        float* dst = (nslabs == 1 && !unpack) ? dW : slabs;
        const int Kp = S ? Mq / S : 0;
        if (S) {
            if (direct) {
                GemmShadows x;
                x.transA = true;
                if (tr_form) { x.A16 = A16; x.B16p = dY16; }
                // otherwise the bias gradient rides along: the kernel's row-tile-0 blocks sum the dY columns they stage anyway
                fused_bias = !tr_form && db != nullptr && (int64_t)(nslabs + 1) * Nout <= t->cs_floats;
                if (fused_bias) { x.colsum = t->cs_ws; x.strideCS = Nout; }
                if (int e = launch_gemm_bf16_x(m->prof, A, Kin, (int64_t)Kp * Kin, dY, Nout, (int64_t)Kp * Nout, dst, Nout,
                                               (int64_t)Kin * Nout, nullptr, nullptr, Kin, Nout, Kp, S, 0, x, s))
                    return e;
            } else {
                if (int e = launch_transpose(A, t->at, Mq, Kin, 1, s)) return e;          // at = (Kin, Mq)
                if (int e = launch_gemm_ex(m->prof, t->at, Mq, S == 1 ? 0 : Kp, dY, Nout, S == 1 ? 0 : (int64_t)Kp * Nout, dst, Nout,
                                           (int64_t)Kin * Nout, nullptr, nullptr, Kin, Nout, Kp, S, 0, s))
                    return e;
            }
        }
        if (R && tr_form) {                      // leftover rows [Mq, M) from the same shadows
            if (int e = launch_dw_tail_bf16(A16 + (int64_t)Mq * Kin, Kin, dY16 + (int64_t)Mq * Nout, Nout, dst + (int64_t)S * Kin * Nout, R, Kin,
                                            Nout, s))
                return e;
        } else if (R) {                          // leftover rows [Mq, M): transposed copy (Kin, R), guarded kernel, slab S
            float* atr = t->at + (int64_t)Kin * Mq;
            if (int e = launch_transpose(A + (int64_t)Mq * Kin, atr, R, Kin, 1, s)) return e;
            if (int e = launch_gemm_ex(m->prof, atr, R, 0, dY + (int64_t)Mq * Nout, Nout, 0, dst + (int64_t)S * Kin * Nout, Nout, 0,
                                       nullptr, nullptr, Kin, Nout, R, 1, 0, s))
                return e;
        }
        // sum the slabs (rows = nslabs, cols = Kin * Nout); the one-chunk partial scratch lives behind the slabs
        if (nslabs > 1 || unpack)
            if (int e = fold_slabs(nslabs)) return e;
        if (fused_bias) {
            // per-slab column sums of dY are in cs_ws (S, Nout); the leftover rows add one more row, then one small fold
            if (R)
                if (int e = launch_colsum(dY + (int64_t)Mq * Nout, t->cs_ws + (int64_t)S * Nout, R, Nout, red_ws, 0, s)) return e;
            if (int e = launch_colsum(t->cs_ws, db, nslabs, Nout, red_ws, 0, s)) return e;
        }
    }
    if (db && !fused_bias) {
        W2V2_REQUIRE(dY, "weight_grad: the bias gradient needs the fp32 dY");
        if (int e = launch_colsum(dY, db, M, Nout, red_ws, 0, s)) return e;
    }
    return W2V2_OK;
}

extern "C" {

int w2v2_set_trainable(w2v2_model* m, const char* prefix, int trainable) {
    W2V2_REQUIRE(m && prefix, "set_trainable: null argument");
    TrainState* t = get_state(m);
    int hits = 0;
    for (size_t i = 0; i < m->params.size(); ++i)
        if (m->params[i].name.compare(0, strlen(prefix), prefix) == 0) {
            t->trainable[i] = trainable ? 1 : 0;
            t->adam_table_fresh = false;
            ++hits;
        }
    if (!hits) {
        set_error("set_trainable: no variable starts with `%s`", prefix);
        return W2V2_ENOTFOUND;
    }
    return W2V2_OK;
}

int w2v2_set_trainable_flags(w2v2_model* m, const uint8_t* flags, int32_t n) {
    W2V2_REQUIRE(m && flags, "set_trainable_flags: null argument");
    W2V2_REQUIRE(n == (int32_t)m->params.size(), "set_trainable_flags: %d flags for %d variables", n, (int)m->params.size());
    TrainState* t = get_state(m);
    bool changed = false;
    for (int32_t i = 0; i < n; ++i) {
        const uint8_t f = flags[i] ? 1 : 0;
        changed |= t->trainable[i] != f;
        t->trainable[i] = f;
    }
    if (changed) t->adam_table_fresh = false;
    return W2V2_OK;
}

int w2v2_train_forward(w2v2_model* m, const float* wave, int32_t B, int64_t L, const int32_t* mask,
                       const uint8_t* spec_mask_host, const float* sd_keep_host, float dropout_p,
                       uint64_t seed, float* logits_out, void* stream) {
    W2V2_REQUIRE(m && wave && logits_out, "train_forward: null argument");
    // ("f16x2" is an inference mode -- its range contract |activation| < 4094 is not watched by the training kernels, and
    //  the step would silently run the six-product bf16x3 path instead)
    W2V2_REQUIRE(m->precision != 3, "train_forward: precision f16x2 is an inference-forward mode; train in fp32, bf16 or bf16x3");
    W2V2_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "train_forward: dropout %f outside [0, 1)", dropout_p);
    // (the keep decisions compare 16 hash bits with floor(p 2^16): a positive p below that resolution would scale by 1 / (1 - p) in
    //  some kernels and not drop at all in others -- one predicate everywhere: every accepted p > 0 has a non-zero threshold)
    W2V2_REQUIRE(dropout_p == 0.f || dropout_p >= 1.0f / 65536.0f, "train_forward: dropout %g is below the 2^-16 resolution of the keep decisions", dropout_p);
    if (!m->finalized) {
        set_error("train_forward: call w2v2_finalize after setting the variables");
        return W2V2_ESTATE;
    }
    const w2v2_config& c = m->cfg;
    PrecisionScope precision(m->precision);
    StepProfScope step_prof(m->prof);          // the training-only kernels' launchers find the profiler here (common.h)
    const int64_t Tll = w2v2_num_frames(m, L);
    W2V2_REQUIRE(Tll >= 1, "train_forward: input too short");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (int e = w2v2_ensure_workspace(m, B, L)) return e;
    const int T = (int)Tll;
    const int H = c.hidden_size, F = c.intermediate_size;
    const int64_t BT = (int64_t)B * T;
    // Precision mode 1: the same bf16 shadows as the inference forward (w2v2_api.hip).  Every producer of a forward GEMM
    // operand -- conv0, GEMM epilogues, LayerNorm, dropout, attention -- also writes the nearest-even bf16 copy, so the
    // forward GEMMs stream 2-byte operands by LDS-DMA instead of converting fp32 in registers.  The shadow buffers are
    // transient scratch (the backward works from the saved fp32 activations); results are bit-identical either way.
    const bool sh = m->precision == 1 && w2v2_shadows_enabled(m);
    if (sh)       // (also refreshes the bf16 weight shadows after an optimizer step: m->w16_valid, which the decisions below read)
        if (int e = w2v2_ensure_shadows(m, B, T, s)) return e;
    const bool attn16 = sh && attention_bf16_supported(H / c.num_heads);
    // FFN hidden activations as bf16 only: needs the shadow paths on both sides (forward GEMM by LDS-DMA, weight gradient in
    // the transposing-read form, whole 128-tiles) and the plain bf16 copies of the FFN kernels for the data gradients
    const bool ffn16_only = sh && m->w16_valid && F % 128 == 0 && H % 128 == 0 && (BT * F) % 4 == 0 && !m->w16p.empty();
    // Precision mode 1 keeps the FFN pre-activation u = t2 W1 + b1 as bf16 (what a mixed_bfloat16 Dense hands to its activation): on
    // the shadow path the up-projection writes ONLY the bf16 copy (its bf16 epilogue; 302 MB of fp32 stores per layer gone at B = 32)
    // and GELU + dropout, GELU' in the backward read that; the shadow-free path of the mode rounds the fp32 u on the way into the
    // same kernels, so both paths agree bit for bit.
    const bool u16_only = ffn16_only && tune_int("W2V2_U16", 1) != 0;
    // The attention output O: its readers are the out-projection GEMM and that GEMM's weight gradient (both stream the bf16 shadow)
    // and D = rowsum(dO o O) of the attention backward, which is defined on the bf16 values -- no fp32 copy is written.
    const bool ctx16_only = ffn16_only && attn16 && tune_int("W2V2_CTX16", 1) != 0;
    // Prenorm: a = LN1(x) and t2 = LN2(t1) are read only by GEMMs that stream their shadows (forward GEMM by LDS-DMA, weight gradient in the
    // transposing-read form: the same conditions as above): written only as bf16 (196 MB of fp32 stores per layer gone at 16 x 480000)
    const bool ln16_only = c.attention_norm_type == 1 && ffn16_only && attn16 && tune_int("W2V2_LN16", 1) != 0;
    // (tools-only build: W2V2_LEAN_WS = 0 allocates everything, as rounds 2-4 did, for A/B runs)
    const int lean = tune_int("W2V2_LEAN_WS", 1) == 0 ? 0
                     : (attn16 ? LEAN_QKV : 0) | (ctx16_only ? LEAN_CTX : 0) | (ffn16_only ? LEAN_GD : 0) | (u16_only ? LEAN_U_HALF : 0) |
                       (ln16_only ? LEAN_LN : 0);
    if (int e = ensure_train_ws(m, B, L, T, lean)) return e;
    TrainState* t = m->train;
    Profiler* pf = m->prof;
    const int act = c.is_gelu_approx ? 2 : 1;
    // element-wise kernels in precision mode 1 evaluate exact GELU / GELU' through the 5-term erf the bf16 GEMM epilogue uses (act 3)
    const int act_ew = (act == 1 && m->precision == 1) ? 3 : act;
    const bool layer_mode = c.feature_extractor_norm_type == 1;
    const float eps = c.layer_norm_eps, p = dropout_p;
    t->p = p;
    t->seed = seed;
    auto fe = [&](int i, const char* leaf) { return m->P("feature_extractor/conv_layers/" + std::to_string(i) + leaf); };

    auto gemm = [&](const float* A, const uint16_t* A16, int64_t lda, int64_t strideA, const float* Bw, int64_t ldb, float* Cc,
                    uint16_t* C16, int64_t ldc, int64_t strideC, const float* bias, const float* res, int M, int N, int K,
                    int nbatch, int act_) -> int {
        if (w2v2_use_split_gemm(m, A, lda, strideA, ldb, M, N, K, nbatch)) {      // precision mode 2 (gemm_split.hip)
            const uint16_t* planes = nullptr;
            if (int e = w2v2_split_planes(m, Bw, K, N, s, &planes)) return e;
            return launch_gemm_split(pf, A, lda, strideA, planes, Cc, ldc, strideC, bias, res, M, N, K, nbatch, act_, s);
        }
        if (!sh) return launch_gemm(pf, A, lda, strideA, Bw, ldb, Cc, ldc, strideC, bias, res, M, N, K, nbatch, act_, s);
        GemmShadows x;
        x.A16 = A16; x.B16 = m->w16[Bw]; x.C16 = C16; x.ldb16 = K;
        return launch_gemm_bf16_x(pf, A, lda, strideA, Bw, ldb, 0, Cc, ldc, strideC, bias, res, M, N, K, nbatch, act_, x, s);
    };
    auto S16 = [&](uint16_t* p16) -> uint16_t* { return sh ? p16 : nullptr; };
    const int NC = c.num_conv_layers;

    // ---- frozen feature extractor: identical to inference (no dropout inside, feature_extractor.py:54-59) ----
    m->acts_skipped.clear();                   // conv outputs written only as bf16: see w2v2_api.hip::w2v2_conv_out_bf16_only
    for (int i = 0; i + 1 < NC; ++i)
        if (w2v2_conv_out_bf16_only(m, i, sh) || w2v2_conv_ln_bf16_only(m, i, sh)) m->acts_skipped.push_back("conv" + std::to_string(i));
    if (int e = launch_conv0_x(pf, wave, fe(0, "/conv/kernel"), c.conv_bias ? fe(0, "/conv/bias") : nullptr,
                               fe(0, "/layer_norm/gamma"), fe(0, "/layer_norm/beta"),
                               (w2v2_conv_out_bf16_only(m, 0, sh) || w2v2_conv_ln_bf16_only(m, 0, sh)) ? nullptr : m->conv[0],
                               sh ? m->conv16[0] : nullptr, m->conv0_ws, B, L, c.kernal_sizes[0], c.strides[0],
                               c.filter_sizes[0], 1e-5f, layer_mode ? 2 : 0, act_ew, s))      // (layer mode: conv + LayerNorm + GELU in one pass)
        return e;
    for (int i = 1; i < NC; ++i) {
        const int cin = c.filter_sizes[i - 1], cout = c.filter_sizes[i];
        const int Tin = m->conv_T[i - 1], Tout = m->conv_T[i];
        uint16_t* o16 = (sh && i + 1 < NC) ? m->conv16[i] : nullptr;
        if (int e = gemm(m->conv[i - 1], sh ? m->conv16[i - 1] : nullptr, (int64_t)c.strides[i] * cin, (int64_t)Tin * cin,
                         fe(i, "/conv/kernel"), cout, w2v2_conv_out_bf16_only(m, i, sh) ? nullptr : m->conv[i], layer_mode ? nullptr : o16, cout,
                         (int64_t)Tout * cout, c.conv_bias ? fe(i, "/conv/bias") : nullptr, nullptr, Tout, cout, c.kernal_sizes[i] * cin, B,
                         layer_mode ? 0 : act))
            return e;
        if (layer_mode)
            if (int e = launch_layer_norm_x(pf, m->conv[i], w2v2_conv_ln_bf16_only(m, i, sh) ? nullptr : m->conv[i], fe(i, "/layer_norm/gamma"), fe(i, "/layer_norm/beta"),
                                            (int64_t)B * Tout, cout, 1e-5f, act_ew, o16, s))
                return e;
    }
    // ---- feature projection: LN -> Dense -> Dropout (feature_extractor.py:92-95) ----
    const int C = c.filter_sizes[NC - 1];
    const float* conv_out = m->conv[NC - 1];
    if (int e = launch_layer_norm_x(pf, conv_out, m->ln512, m->P("feature_projection/layer_norm/gamma"),
                                    m->P("feature_projection/layer_norm/beta"), BT, C, eps, 0, S16(m->ln512_16), s))
        return e;
    if (int e = gemm(m->ln512, S16(m->ln512_16), C, 0, m->P("feature_projection/projection/kernel"), H, m->proj, nullptr, H, 0,
                     m->P("feature_projection/projection/bias"), nullptr, (int)BT, H, C, 1, 0))
        return e;
    if (int e = launch_dropout_fwd(m->proj, nullptr, t->hd, BT * H, 0, p, seed, DS_FEATURE_PROJECTION, s)) return e;
    // ---- spec-augment: masked frames <- masked_spec_embed (modeling.py:193-199, spec_augment.py:119-127) ----
    t->have_spec = spec_mask_host != nullptr;
    const float* enc_x = t->hd;
    if (t->have_spec) {
        W2V2_HIP_CHECK(hipMemcpyAsync(t->spec_mask, spec_mask_host, (size_t)BT, hipMemcpyHostToDevice, s));
        if (int e = launch_spec_aug_fwd(t->hd, t->spec_mask, m->P("masked_spec_embed"), t->hm, BT, H, s)) return e;
        enc_x = t->hm;
    }
    // ---- encoder (encoder.py:251-276) ----
    const int32_t* flen = nullptr;
    t->have_mask = mask != nullptr;
    if (mask) {
        if (int e = launch_frame_lengths(pf, mask, m->frame_len, B, L, c.kernal_sizes, c.strides, c.num_conv_layers, s)) return e;
        flen = m->frame_len;
    }
    // posout = xz + GELU(c), c saved for backward
    if (w2v2_pos_conv_bf16_ok(m)) {      // precision mode 1: batched bf16 GEMM (posconv.hip); m->t0 is free scratch here
        if (int e = w2v2_ensure_pos16(m, B, T, s)) return e;
        if (int e = launch_pos_conv_bf16(pf, enc_x, m->pos_w16, m->P("encoder/pos_conv_embed/conv/bias"), flen, m->posout, t->pos_c,
                                         m->pos_pack16, m->t0, B, T, H, c.num_conv_pos_embeddings, c.num_conv_pos_embedding_groups,
                                         act, c.num_conv_pos_embeddings / 2, 1, s))
            return e;
    } else if (int e = launch_pos_conv_ex(pf, enc_x, m->pos_wg, m->P("encoder/pos_conv_embed/conv/bias"), flen, m->posout, t->pos_c,
                                          B, T, H, c.num_conv_pos_embeddings, c.num_conv_pos_embedding_groups, act,
                                          c.num_conv_pos_embeddings / 2, 1, s)) {
        return e;
    }
    const bool prenorm = c.attention_norm_type == 1;
    // postnorm: hs[0] = dropout(LN(posout));  prenorm: hs[0] = dropout(posout)   (encoder.py:267-270)
    {
        const float* pre = m->posout;
        if (!prenorm) {
            if (int e = launch_layer_norm(pf, m->posout, m->t0, m->P("encoder/layer_norm/gamma"), m->P("encoder/layer_norm/beta"), BT, H, eps, 0, s)) return e;
            pre = m->t0;
        }
        float* h0 = prenorm ? t->hs0 : m->hs[0];      // prenorm: m->hs[0] aliases posout, keep the dropped copy apart
        // postnorm: layer 0's q|k|v GEMM reads this tensor -> shadow
        if (int e = launch_dropout_fwd_x(pre, nullptr, h0, (sh && !prenorm) ? m->hs16[0] : nullptr, BT * H, 0, p, seed, DS_ENCODER_IN, s)) return e;
    }
    t->ffn16_only = ffn16_only;       // (decided in front of the workspace, which leaves out the fp32 buffers these modes never write)
    t->u16_only = u16_only;
    t->ctx16_only = ctx16_only;
    t->ln16_only = ln16_only;
    const int u_round = m->precision == 1 ? 1 : 0;
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string b = "encoder/layers/" + std::to_string(i);
        LayerSave& l = t->layers[i];
        const float* x = (prenorm && i == 0) ? t->hs0 : m->hs[i];
        l.keep = sd_keep_host ? sd_keep_host[i] : 1.0f;
        const float* attn_in = x;
        const uint16_t* attn_in16 = (sh && !prenorm) ? m->hs16[i] : nullptr;   // postnorm: written by the producer of hs[i]
        if (prenorm) {     // x + drop(attn(LN(x)))   (encoder.py:114-119)
            if (int e = launch_layer_norm_x(pf, x, ln16_only ? nullptr : l.a, m->P(b + "/layer_norm/gamma"), m->P(b + "/layer_norm/beta"), BT, H, eps, 0,
                                            S16(l.a16), s))
                return e;
            attn_in = ln16_only ? nullptr : l.a;
            attn_in16 = S16(l.a16);
        }
        // the bf16 attention kernels (forward and backward) read q | k | v only as bf16: the projection then writes just that shadow
        if (int e = gemm(attn_in, attn_in16, H, 0, m->qkv_w[i], 3 * H, attn16 ? nullptr : l.qkv, attn16 ? l.qkv16 : nullptr, 3 * H, 0, m->qkv_b[i],
                         nullptr, (int)BT, 3 * H, H, 1, 0))
            return e;
        AttnTrain tr{p, seed, layer_stream(i, 0), l.lse, attn16 ? l.keep_bits : nullptr};
        if (int e = launch_attention_train_x(pf, attn16 ? nullptr : l.qkv, attn16 ? l.qkv16 : nullptr, flen, ctx16_only ? nullptr : l.ctx, attn16 ? l.ctx16 : nullptr, B, T,
                                             H, c.num_heads, tr, s))
            return e;
        // o = ctx Wo + bo;  t1 = dropout(o) + x   (encoder.py:116-119)
        // (dropout + residual as their own pass: folded into this GEMM's epilogue they were measured slower, study section 12 / r03)
        if (int e = gemm(ctx16_only ? nullptr : l.ctx, attn16 ? l.ctx16 : nullptr, H, 0, m->P(b + "/attention/out_proj/kernel"), H, m->t0, nullptr, H, 0,
                         m->P(b + "/attention/out_proj/bias"), nullptr, (int)BT, H, H, 1, 0))
            return e;
        // postnorm: t2 = LN1(t1) feeds the FFN and is its residual; prenorm: t2 = LN2(t1) feeds the FFN, t1 is the residual
        const char* ln_a = prenorm ? "/final_layer_norm" : "/layer_norm";
        // (round 4: dropout + residual + LayerNorm as one pass over the row -- t1 is written once and not read back)
        // (the fused pass takes rows of up to 2048 channels, a multiple of 4, 16-byte aligned: anything else falls back to the two kernels)
        if (H % 4 == 0 && H <= 2048 && tune_int("W2V2_LN_DROP", 1) != 0 &&
            ((reinterpret_cast<uintptr_t>(m->t0) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(l.t1) | reinterpret_cast<uintptr_t>(l.t2) |
              reinterpret_cast<uintptr_t>(m->P(b + ln_a + "/gamma")) | reinterpret_cast<uintptr_t>(m->P(b + ln_a + "/beta"))) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(S16(l.t2_16)) & 7) == 0) {
            if (int e = launch_layer_norm_drop(pf, m->t0, x, l.t1, ln16_only ? nullptr : l.t2, S16(l.t2_16), m->P(b + ln_a + "/gamma"), m->P(b + ln_a + "/beta"), BT, H, eps, p,
                                               seed, layer_stream(i, 1), s))
                return e;
        } else {
            if (int e = launch_dropout_fwd(m->t0, x, l.t1, BT * H, 0, p, seed, layer_stream(i, 1), s)) return e;
            if (int e = launch_layer_norm_x(pf, l.t1, ln16_only ? nullptr : l.t2, m->P(b + ln_a + "/gamma"), m->P(b + ln_a + "/beta"), BT, H, eps, 0, S16(l.t2_16), s)) return e;
        }
        const float* ffn_res = prenorm ? l.t1 : l.t2;
        float* ffn_out = prenorm ? m->hs[i + 1] : l.t3;
        if (l.keep != 0.f) {
            // u = t2 W1 + b1;  gd = dropout(GELU(u));  out = res + keep * (gd W2 + b2)   (encoder.py:127-130)
            // (ffn16_only: every reader of gd -- the next GEMM, and the down-projection's weight gradient -- streams the bf16 shadow)
            uint16_t* const u16 = u16_only ? reinterpret_cast<uint16_t*>(l.u) : nullptr;
            if (int e = gemm(ln16_only ? nullptr : l.t2, S16(l.t2_16), H, 0, m->P(b + "/feed_forward/intermediate_dense/kernel"), F, u16_only ? nullptr : l.u, u16, F, 0,
                             m->P(b + "/feed_forward/intermediate_dense/bias"), nullptr, (int)BT, F, H, 1, 0))
                return e;
            EwBf16 uin;
            uin.a16 = u16; uin.round_in = u_round;
            if (int e = launch_dropout_fwd_x(u16_only ? nullptr : l.u, nullptr, ffn16_only ? nullptr : l.gd, S16(l.gd16), BT * F, act_ew, p, seed,
                                             layer_stream(i, 2), s, uin))
                return e;
            if (int e = gemm(ffn16_only ? nullptr : l.gd, S16(l.gd16), F, 0, m->P(b + "/feed_forward/output_dense/kernel"), H, ffn_out, nullptr, H, 0,
                             m->P(b + "/feed_forward/output_dense/bias"), ffn_res, (int)BT, H, F, 1, 0))
                return e;
        } else {
            W2V2_HIP_CHECK(hipMemcpyAsync(ffn_out, ffn_res, (size_t)BT * H * 4, hipMemcpyDeviceToDevice, s));
        }
        if (!prenorm)
            if (int e = launch_layer_norm_x(pf, l.t3, m->hs[i + 1], m->P(b + "/final_layer_norm/gamma"),
                                            m->P(b + "/final_layer_norm/beta"), BT, H, eps, 0, S16(m->hs16[i + 1]), s))
                return e;
    }
    // ---- [prenorm: final encoder LN] -> Dropout -> lm_head (encoder.py:274-275, modeling.py:253-254) ----
    const float* head_in = m->hs[c.num_layers];
    if (prenorm) {
        if (int e = launch_layer_norm(pf, m->hs[c.num_layers], m->enc_out, m->P("encoder/layer_norm/gamma"),
                                      m->P("encoder/layer_norm/beta"), BT, H, eps, 0, s))
            return e;
        head_in = m->enc_out;
    }
    if (!c.with_lm_head) {
        // Wav2Vec2Model.call(training=True) (modeling.py:169-209): the backbone's hidden states (B, T, H); the head's Dropout and
        // Dense belong to Wav2Vec2ForCTC.  No backward from here (the training step differentiates the CTC model).
        W2V2_HIP_CHECK(hipMemcpyAsync(logits_out, head_in, (size_t)BT * H * 4, hipMemcpyDeviceToDevice, s));
        t->forward_done = false;
        return W2V2_OK;
    }
    if (int e = launch_dropout_fwd_x(head_in, nullptr, t->hdf, S16(m->enc16), BT * H, 0, p, seed, DS_HEAD, s)) return e;
    if (int e = gemm(t->hdf, S16(m->enc16), H, 0, m->P("lm_head/kernel"), c.vocab_size, logits_out, nullptr, c.vocab_size, 0,
                     m->P("lm_head/bias"), nullptr, (int)BT, c.vocab_size, H, 1, 0))
        return e;
    t->forward_done = true;
    t->x16_valid = sh;                 // the per-layer X shadows (and hs16) hold this forward's activations
    t->x16_attn = attn16;
    return W2V2_OK;
}

int w2v2_train_backward(w2v2_model* m, const float* dlogits, void* stream) {
    W2V2_REQUIRE(m && dlogits, "train_backward: null argument");
    TrainState* t = m->train;
    if (!t || !t->forward_done) {
        set_error("train_backward: no training forward to differentiate");
        return W2V2_ESTATE;
    }
    const w2v2_config& c = m->cfg;
    PrecisionScope precision(m->precision);
    StepProfScope step_prof(m->prof);          // the training-only kernels' launchers find the profiler here (common.h)
    // dX = dY W^T reads the fp32 transposed copy WT ([out][in]) as its B operand; in precision mode 1 with shadows the bf16
    // copy of W itself ([in][out] = (N, K) for this GEMM) is the B shadow -- no transpose needed.  Bit-identical results.
    const bool shb = m->precision == 1 && w2v2_shadows_enabled(m) && m->w16_valid;
    // A16: the producer's bf16 shadow of A (or null): with it both operands stream by LDS-DMA (gemm_bf16.hip source 5)
    auto gemm_dx = [&](const float* A, const uint16_t* A16, int64_t lda, const float* WT, const float* W, float* Cc, int64_t ldc,
                       const float* res, int M, int N, int K, hipStream_t st, uint16_t* C16 = nullptr) -> int {
        // (C16: bf16 shadow of the result, only from the shadow branch -- callers ask for it only when `dx_shadowed(W)`)
        if (shb) {
            auto it = m->w16p.find(W);
            if (it != m->w16p.end()) {
                GemmShadows x;
                x.A16 = A16;
                x.C16 = C16;
                x.B16 = it->second;
                x.ldb16 = K;
                return launch_gemm_bf16_x(m->prof, A, lda, 0, WT, N, 0, Cc, ldc, 0, nullptr, res, M, N, K, 1, 0, x, st);
            }
        }
        if (w2v2_use_split_gemm(m, A, lda, 0, N, M, N, K, 1)) {                   // precision mode 2: planes of the transposed copy
            const uint16_t* planes = nullptr;
            if (int e = w2v2_split_planes(m, WT, K, N, st, &planes)) return e;
            return launch_gemm_split(m->prof, A, lda, 0, planes, Cc, ldc, 0, nullptr, res, M, N, K, 1, 0, st);
        }
        return launch_gemm(m->prof, A, lda, 0, WT, N, Cc, ldc, 0, nullptr, res, M, N, K, 1, 0, st);
    };
    auto dx_shadowed = [&](const float* W) { return shb && m->w16p.find(W) != m->w16p.end(); };
    // bf16 shadows of dY (written by its producer) for the data-gradient GEMMs: only where the consumer will take them
    const int dhead = c.hidden_size / c.num_heads;
    uint16_t* const s16h = shb ? t->dy16_h : nullptr;
    uint16_t* const s16f = shb ? t->dy16_f : nullptr;
    uint16_t* const s16q = (shb && attention_bf16_supported(dhead)) ? t->dy16_3h : nullptr;
    if (shb && ((int64_t)t->B * t->T) % 64 != 0) {
        // Ragged row count: the weight-gradient GEMMs read the missing rows of their last K tile from the zero row behind each dY
        // shadow.  Re-assert the invariant on THIS stream every backward (three rows, a few KB): nothing may have padded into them.
        hipStream_t st0 = reinterpret_cast<hipStream_t>(stream);
        const int64_t BT0 = (int64_t)t->B * t->T;
        W2V2_HIP_CHECK(hipMemsetAsync(t->dy16_h + BT0 * c.hidden_size, 0, (size_t)c.hidden_size * 2, st0));
        W2V2_HIP_CHECK(hipMemsetAsync(t->dy16_f + BT0 * c.intermediate_size, 0, (size_t)c.intermediate_size * 2, st0));
        W2V2_HIP_CHECK(hipMemsetAsync(t->dy16_3h + BT0 * 3 * c.hidden_size, 0, (size_t)3 * c.hidden_size * 2, st0));
    }
    const bool xs = shb && t->x16_valid;             // X shadows of the forward are there for the weight-gradient GEMMs
    // the forward kept the FFN hidden activations only as bf16: the backward must then run entirely on the shadow paths
    const bool f16 = t->ffn16_only;
    W2V2_REQUIRE(!f16 || (xs && t->dy16_f && t->dy16_h), "train_backward: the forward ran with bf16 shadows (FFN activations kept as bf16 only); "
                                                      "precision / W2V2_BF16_SHADOWS must not change before the backward");
    for (size_t i = 0; i < m->params.size(); ++i)
        if (t->trainable[i] && m->params[i].name.compare(0, 18, "feature_extractor/") == 0) {
            set_error("train_backward: `%s` is trainable, but the conv feature extractor has no backward "
                      "(the reference freezes it, main.py:234-237); call freeze_feature_extractor()", m->params[i].name.c_str());
            return W2V2_ESTATE;
        }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Profiler* pf = m->prof;
    const int B = t->B, T = t->T, H = c.hidden_size, F = c.intermediate_size, V = c.vocab_size;
    const int64_t BT = (int64_t)B * T;
    const int act = c.is_gelu_approx ? 2 : 1;
    // element-wise kernels in precision mode 1 evaluate exact GELU / GELU' through the 5-term erf the bf16 GEMM epilogue uses (act 3)
    const int act_ew = (act == 1 && m->precision == 1) ? 3 : act;
    const float eps = c.layer_norm_eps, p = t->p;
    const uint64_t seed = t->seed;
    const int32_t* flen = t->have_mask ? m->frame_len : nullptr;
    // With the shadows on, the weight-gradient GEMMs take the LDS-DMA / transposing-read kernel, which has no fp32 dY in
    // registers to sum for the bias gradient: the PRODUCER of each dY leaves its column sums instead (dropout backward,
    // LayerNorm backward), and weight_grad is called without a bias target.  `bias_from_producer` says that happened.
    // Deferred folds of the encoder layers (train.h: FoldBatch; W2V2_OPT_DEFER_FOLDS): `fb` is set below, once the side-stream decision is known
    FoldBatch fold;
    FoldBatch* fb = nullptr;
    auto dropout_bwd_bias = [&](const float* u, const float* dy, float* dx, uint16_t* dx16, int64_t rows, int cols, int act_, uint32_t stream_id,
                                float* bias_grad, bool* bias_done, const EwBf16& in = EwBf16{}, bool deferrable = false) -> int {
        *bias_done = false;
        if (shb && bias_grad) {
            *bias_done = true;
            const bool d = deferrable && fb;        // (the FFN site: its partial rows live in site_ws[1] until the layer's fold launch)
            return launch_dropout_bwd_colsum(u, dy, dx, dx16, bias_grad, rows, cols, act_, p, seed, stream_id, d ? t->site_ws[1] : t->red_ws, s, in,
                                             d ? fb : nullptr);
        }
        return launch_dropout_bwd_x(u, dy, dx, dx16, rows * cols, act_, p, seed, stream_id, s, in);
    };
    // Gradient buffer at the start of a backward.  Rounds 1-5 zeroed all of it every step (360 MB / 1.25 GB: the runtime's fill kernels,
    // 0.23 / 0.8 ms per step and the only framework kernels left in it).  Every producer below STORES its gradient (no accumulation
    // into the buffer: folds, column sums and GEMM epilogues overwrite), so a slot needs zeros only if nothing will write it this time:
    //   * slots of variables that are not trainable, and the 16-byte padding between slots: zero since allocation, never written;
    //   * masked_spec_embed when this step has no spec-augment mask;
    //   * everything, conservatively, when the trainable set changed since the last backward (a slot that was written before keeps
    //     its last gradient) or when stochastic depth dropped a layer (its branch's gradients are not computed: zeros).
    // tests/test_train_gpu.py poisons the buffer with NaN before its gradient checks: a trainable slot nobody wrote shows up there.
    {
        uint64_t sig = 1469598103934665603ull;
        for (size_t i = 0; i < t->trainable.size(); ++i) sig = (sig ^ (uint64_t)(t->trainable[i] ? 0x9E : 0x3C)) * 1099511628211ull;
        bool dropped = false;
        for (int i = 0; i < c.num_layers; ++i) dropped = dropped || t->layers[i].keep == 0.f;
        if (sig != t->trainable_sig || dropped || t->grads_need_clear) {
            W2V2_HIP_CHECK(hipMemsetAsync(t->grads, 0, (size_t)t->gtotal * 4, s));
            t->trainable_sig = sig;
            t->grads_need_clear = dropped;        // (the step after a dropped layer starts from zeros as well: its slots hold nothing new)
        } else if (!t->have_spec) {
            auto it = m->index.find("masked_spec_embed");
            if (it != m->index.end() && t->trainable[it->second])
                W2V2_HIP_CHECK(hipMemsetAsync(t->grads + t->goff[it->second], 0, (size_t)m->params[it->second].numel * 4, s));
        }
    }
    if (int e = refresh_transposes(m, s)) return e;
    const int nbuckets = c.num_layers + 2;
    while ((int)t->bucket_ev.size() < nbuckets) {
        hipEvent_t ev;
        W2V2_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        t->bucket_ev.push_back(ev);
    }
    // ---- weight-gradient side stream (TrainState::wg_stream): the bf16 shadow path only -- there every dY a weight gradient reads is a
    // shadow with a known next writer (the wait points below); the other paths keep one stream.
    // (tools-only build: W2V2_WGRAD_STREAM = 0 / 1 overrides the option for A/B runs)
    const int wg_knob = tune_int("W2V2_WGRAD_STREAM", -1);
    const bool side_on = shb && t->x16_valid && t->x16_attn && t->attn_colpart && (wg_knob < 0 ? m->opt_wgrad_stream : wg_knob != 0);
    if (side_on && !t->wg_stream) {
        int lo = 0, hi = 0;
        W2V2_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));           // (lo = the numerically largest = least urgent)
        W2V2_HIP_CHECK(hipStreamCreateWithPriority(&t->wg_stream, hipStreamNonBlocking, lo));
        W2V2_HIP_CHECK(hipEventCreateWithFlags(&t->wg_fork, hipEventDisableTiming));
        W2V2_HIP_CHECK(hipEventCreateWithFlags(&t->wg_main, hipEventDisableTiming));
        for (int k = 0; k < 4; ++k) W2V2_HIP_CHECK(hipEventCreateWithFlags(&t->wg_join[k], hipEventDisableTiming));
    }
    // (the side stream keeps the folds behind their GEMMs: its weight gradients run ahead of / behind the main stream's position)
    // (tools-only build: W2V2_DEFER_FOLDS = 0 none, 1 the column-sum folds only -- slab sums behind their GEMMs --, 2 everything.
    //  One box, arms interleaved, profiles/r05_ab_defer_folds.txt: base step 33.80 / 33.55 / 33.50 ms, large-robust 98.38 / 97.94 / 97.74)
    const int defer_mode = (!side_on && m->opt_defer_folds) ? tune_int("W2V2_DEFER_FOLDS", 2) : 0;
    if (defer_mode != 0) fb = &fold;
    if (int e = ensure_site_scratch(m, side_on, defer_mode != 0, defer_mode == 2)) return e;
    auto wg_site = [&](int k) {
        WgSite w;
        if (defer_mode == 2) { w.slabs = t->site_slabs[k]; w.slab_floats = t->site_slab_floats[k]; w.defer = fb; }
        return w;
    };
    float* const ln2_ws = fb ? t->site_ws[0] : t->red_ws;
    float* const ln1_ws = fb ? t->site_ws[2] : t->red_ws;
    bool wg_pending[4] = {false, false, false, false};
    bool wg_used = false;
    // side(site, f): run f(stream, reduction scratch) -- a weight gradient -- behind everything the main stream has enqueued so far
    auto side = [&](int site, auto&& f) -> int {
        if (!side_on) return f(s, t->red_ws);
        W2V2_HIP_CHECK(hipEventRecord(t->wg_fork, s));
        W2V2_HIP_CHECK(hipStreamWaitEvent(t->wg_stream, t->wg_fork, 0));
        if (int e = f(t->wg_stream, t->red_ws_side)) return e;
        W2V2_HIP_CHECK(hipEventRecord(t->wg_join[site], t->wg_stream));
        wg_pending[site] = wg_used = true;
        return W2V2_OK;
    };
    // the main stream is about to overwrite what site's weight gradient reads
    auto side_wait = [&](int site) -> int {
        if (wg_pending[site]) {
            W2V2_HIP_CHECK(hipStreamWaitEvent(s, t->wg_join[site], 0));
            wg_pending[site] = false;
        }
        return W2V2_OK;
    };
    // bucket k's slice of the flat buffer is final once this is recorded (w2v2_train_bucket_wait): with the side stream in use the
    // event goes THERE, after the side stream has caught up with the main stream's position (LayerNorm / bias gradients come from it)
    auto bucket_done = [&](int k) -> int {
        if (wg_used) {
            W2V2_HIP_CHECK(hipEventRecord(t->wg_main, s));
            W2V2_HIP_CHECK(hipStreamWaitEvent(t->wg_stream, t->wg_main, 0));
            W2V2_HIP_CHECK(hipEventRecord(t->bucket_ev[k], t->wg_stream));
            return W2V2_OK;
        }
        W2V2_HIP_CHECK(hipEventRecord(t->bucket_ev[k], s));
        return W2V2_OK;
    };
    auto G = [&](const std::string& n) { return is_trainable(m, n) ? grad_of(m, n) : nullptr; };
    // does anything below the head train?
    bool below_head = false;
    for (size_t i = 0; i < m->params.size(); ++i)
        if (t->trainable[i] && m->params[i].name.compare(0, 8, "lm_head/") != 0) below_head = true;

    // ---- head ----
    if (int e = weight_grad(m, t->hdf, dlogits, (int)BT, H, V, G("lm_head/kernel"), G("lm_head/bias"), s)) return e;
    if (int e = bucket_done(0)) return e;
    if (!below_head) {                          // stage 1 of the reference: only lm_head trains (main.py:210)
        for (int k = 1; k < nbuckets; ++k)
            if (int e = bucket_done(k)) return e;
        return W2V2_OK;
    }
    float *dh = t->gh[0], *tmp = t->gh[1], *tmp2 = t->gh[2], *tmp3 = t->gh[3];
    const bool prenorm = c.attention_norm_type == 1;
    float* head_b2 = nullptr;      // prenorm: the encoder LayerNorm's backward already summed dh's columns into the top layer's b2 gradient
    bool head_dh16 = false;        // ... and wrote dh's bf16 shadow
    if (int e = launch_gemm(pf, dlogits, V, 0, t->WlmT, H, tmp, H, 0, nullptr, nullptr, (int)BT, H, V, 1, 0, s)) return e;
    if (!prenorm) {
        if (int e = launch_dropout_bwd(nullptr, tmp, dh, BT * H, 0, p, seed, DS_HEAD, s)) return e;
    } else {   // hdf = dropout(LN_enc(hs[N]))   (encoder.py:274-275)
        if (int e = launch_dropout_bwd(nullptr, tmp, tmp2, BT * H, 0, p, seed, DS_HEAD, s)) return e;
        float* dg = G("encoder/layer_norm/gamma");
        float* db = G("encoder/layer_norm/beta");
        // (round 5: this pass also leaves dh's bf16 shadow -- a separate rounding kernel before -- and, on the shadow path, the column
        //  sums of dh = the top layer's down-projection bias gradient: see `b2_done` in the prenorm loop)
        const bool top16 = s16h && (BT * H) % 4 == 0 && (reinterpret_cast<uintptr_t>(dh) & 15) == 0;
        head_b2 = (shb && tune_int("W2V2_B2_FROM_LN", 1) != 0 && c.num_layers > 0 && t->layers[c.num_layers - 1].keep != 0.f)
                      ? G("encoder/layers/" + std::to_string(c.num_layers - 1) + "/feed_forward/output_dense/bias") : nullptr;
        if (int e = launch_ln_bwd_x(m->hs[c.num_layers], m->P("encoder/layer_norm/gamma"), tmp2, dh, top16 ? s16h : nullptr, dg ? dg : t->dummy,
                                    db ? db : t->dummy + H, BT, H, eps, t->red_ws, s, head_b2))
            return e;
        head_dh16 = top16;
    }

    // dqkv only as bf16 (+ per-block column sums for the bias gradient) when all three of its readers can do without the fp32 copy
    auto dqkv16_only = [&](int i, const uint16_t* attn_in16) {
        return xs && s16q && t->x16_attn && t->attn_colpart && attn_in16 && H % 128 == 0 && dx_shadowed(m->qkv_w[i]);
    };
    auto qkv_weight_grad = [&](const std::string& b, const float* attn_in, const uint16_t* attn_in16, bool only16, hipStream_t s, float* rws) -> int {
        // packed q|k|v projection: dW (H, 3H) -> the three (H, H) kernels, db (3H) -> the three biases
        float* dWqkv = t->dwqkv;
        float* dbqkv = dWqkv + (int64_t)3 * H * H;
        const char* names[3] = {"q_proj", "k_proj", "v_proj"};
        float* gw3[3];
        float* gb3[3];
        bool all6 = true;
        for (int j = 0; j < 3; ++j) {
            gw3[j] = G(b + "/attention/" + names[j] + "/kernel");
            gb3[j] = G(b + "/attention/" + names[j] + "/bias");
            all6 = all6 && gw3[j] && gb3[j];
        }
        if (fb && only16 && all6 && H % 4 == 0 && fold.tab.n + 2 <= FOLD_MAX_JOBS) {
            // deferred: the layer's fold launch sums the (H, 3H) slabs straight into the three (H, H) kernels and the attention backward's
            // column partials into the three biases (group g of the wide job = columns [g H, (g + 1) H) of the 3H-wide partial rows)
            WgSite site = wg_site(3);
            site.unpackH = H;
            for (int j = 0; j < 3; ++j) site.unpack3[j] = gw3[j];
            if (int e = weight_grad(m, attn_in, nullptr, (int)BT, H, 3 * H, dWqkv, nullptr, s, (xs && s16q) ? attn_in16 : nullptr, s16q, s16q != nullptr,
                                    rws, &site))
                return e;
            W2V2_REQUIRE(fb->add_wide(t->attn_colpart, attention_colpart_rows(B, T), H, (int64_t)3 * H, 3, gb3[0], gb3[1], gb3[2]),
                         "train_backward: fold table full");
            return W2V2_OK;
        }
        if (int e = weight_grad(m, attn_in, only16 ? nullptr : t->g3h, (int)BT, H, 3 * H, dWqkv, only16 ? nullptr : dbqkv, s,
                                (xs && s16q) ? attn_in16 : nullptr, s16q, s16q != nullptr, rws))
            return e;
        if (only16)
            if (int e = launch_colsum_fold(t->attn_colpart, dbqkv, attention_colpart_rows(B, T), 3 * H, s)) return e;
        if (H % 4 == 0) return launch_qkv_unpack(dWqkv, dbqkv, gw3, gb3, H, s);
        for (int j = 0; j < 3; ++j) {
            float* gw = G(b + "/attention/" + names[j] + "/kernel");
            float* gb = G(b + "/attention/" + names[j] + "/bias");
            if (gw)
                W2V2_HIP_CHECK(hipMemcpy2DAsync(gw, (size_t)H * 4, dWqkv + j * H, (size_t)3 * H * 4, (size_t)H * 4, (size_t)H,
                                                hipMemcpyDeviceToDevice, s));
            if (gb) W2V2_HIP_CHECK(hipMemcpyAsync(gb, dbqkv + j * H, (size_t)H * 4, hipMemcpyDeviceToDevice, s));
        }
        return W2V2_OK;
    };

    // dgd = dY W2^T (the down-projection's data gradient) into t->gf, then du = dropout-backward(dgd) * GELU'(u) with its column sums
    // (the up-projection's bias gradient) as one element-wise pass.  (In round 2 that pass rode in the GEMM's epilogue; with the GEMM
    // on the 128 x 256 kernel the separate pass at the HBM roofline is faster: profiles/r03_gemm_bf16_study.md.)
    auto ffn_hidden_grad = [&](int i, LayerSave& l, const std::string& b, const float* dy, const uint16_t* dy16, bool du16_only, float* gb1,
                               bool* b1_done) -> int {
        const float* W2 = m->P(b + "/feed_forward/output_dense/kernel");
        *b1_done = false;
        EwBf16 in;
        in.round_in = m->precision == 1 ? 1 : 0;
        if (t->u16_only) {
            // the forward kept u only as bf16; the gradient of the hidden activation is written only as bf16 too (the data-gradient
            // GEMM's bf16 epilogue, into the buffer du's shadow will occupy) and GELU' x dropout-backward runs on it in place
            W2V2_REQUIRE(s16f && dx_shadowed(W2), "train_backward: the forward kept u as bf16 only, which needs the shadow path for the down-projection's data gradient");
            if (int e = gemm_dx(dy, dy16, H, l.W2T, W2, nullptr, F, nullptr, (int)BT, F, H, s, s16f)) return e;
            in.a16 = reinterpret_cast<const uint16_t*>(l.u); in.b16 = s16f;
            return dropout_bwd_bias(nullptr, nullptr, du16_only ? nullptr : t->gf, s16f, BT, F, act_ew, layer_stream(i, 2), gb1, b1_done, in, true);
        }
        if (int e = gemm_dx(dy, dy16, H, l.W2T, W2, t->gf, F, nullptr, (int)BT, F, H, s)) return e;
        return dropout_bwd_bias(l.u, t->gf, du16_only ? nullptr : t->gf, s16f, BT, F, act_ew, layer_stream(i, 2), gb1, b1_done, in, true);
    };
    const bool fuse_do_tail = !tune_int("W2V2_NO_DO_TAIL", 0);
    bool dh16_valid = head_dh16;
    // b2_done: the kernel that produced dh (the encoder LayerNorm's backward for the top layer, the LayerNorm-1 backward of layer i + 1
    // below) also left its column sums -- dh is the dY of this layer's down-projection, so they are its bias gradient -- and the
    // weight gradient needs no separate pass over the fp32 dh (two launches and a 98 MB read per layer at 16 x 480000)
    bool b2_done = head_b2 != nullptr;
    if (prenorm && !dh16_valid && s16h && (BT * H) % 4 == 0 && (reinterpret_cast<uintptr_t>(dh) & 15) == 0) {
        // the last layer's output gradient arrives in fp32 only: round it once so that its down-projection GEMMs stream shadows too
        if (int e = launch_to_bf16(dh, s16h, BT * H, s)) return e;
        dh16_valid = true;
    }
    for (int i = c.num_layers - 1; i >= 0 && prenorm; --i) {
        // prenorm layer (encoder.py:111-134):  t1 = x + drop(attn(LN1(x)));  out = t1 + keep * FFN(LN2(t1))
        const std::string b = "encoder/layers/" + std::to_string(i);
        LayerSave& l = t->layers[i];
        const WgSite ws0 = wg_site(0), ws1 = wg_site(1), ws2 = wg_site(2);
        const float* x = i == 0 ? t->hs0 : m->hs[i];
        float* dt1 = tmp3;
        // dh's bf16 shadow (written by the previous iteration's closing axpby into s16h, which is free again by then)
        const uint16_t* dh16 = dh16_valid ? s16h : nullptr;
        bool bo_done = false;
        float* const gbo = G(b + "/attention/out_proj/bias");
        // (d_o is read only as bf16 when both of its GEMMs stream shadows and its column sums come from the producer: the
        //  dropout backward that makes it then rides in the tail of the LayerNorm backward that produces dt1)
        const bool do16_only = xs && t->x16_attn && s16h && H % 128 == 0 && (BT * H) % 4 == 0 && dx_shadowed(m->P(b + "/attention/out_proj/kernel"));
        bool do_tail = do16_only && shb && fuse_do_tail;
        const LnDropTail do_drop{p, seed, layer_stream(i, 1)};
        if (l.keep != 0.f) {
            W2V2_REQUIRE(!f16 || dh16, "train_backward: no bf16 shadow of the layer's output gradient");
            if (int e = side(0, [&](hipStream_t st, float* rws) {
                    return weight_grad(m, f16 ? nullptr : l.gd, dh, (int)BT, F, H, G(b + "/feed_forward/output_dense/kernel"),
                                       b2_done ? nullptr : G(b + "/feed_forward/output_dense/bias"), st, (xs && dh16) ? l.gd16 : nullptr, dh16,
                                       dh16 != nullptr && dh16 == s16h, rws, &ws0);
                }))
                return e;
            bool b1_done = false;
            float* const gb1 = G(b + "/feed_forward/intermediate_dense/bias");
            // (f16: du is needed only as bf16 -- both consumers stream the shadow -- unless its fp32 column sums are still to be taken)
            const bool du16_only = f16 && (!gb1 || shb) && dx_shadowed(m->P(b + "/feed_forward/intermediate_dense/kernel"));
            float* const du = du16_only ? nullptr : t->gf;
            if (int e = side_wait(1)) return e;              // (du / its shadow: read by the layer above's up-projection weight gradient)
            if (int e = ffn_hidden_grad(i, l, b, dh, dh16, du16_only, gb1, &b1_done)) return e;
            if (int e = side(1, [&](hipStream_t st, float* rws) {
                    return weight_grad(m, l.t2, du, (int)BT, H, F, G(b + "/feed_forward/intermediate_dense/kernel"),
                                       b1_done ? nullptr : gb1, st, xs ? l.t2_16 : nullptr, s16f, s16f != nullptr, rws, &ws1);
                }))
                return e;
            if (int e = gemm_dx(du, s16f, F, l.W1T, m->P(b + "/feed_forward/intermediate_dense/kernel"), tmp, H, nullptr, (int)BT, H, F, s)) return e;
            float* dg2 = G(b + "/final_layer_norm/gamma");
            float* db2 = G(b + "/final_layer_norm/beta");
            // dt1 = dh (residual) + LN2-backward(tmp), one pass (+ d_o's shadow and column sums where nothing reads d_o in fp32;
            // dh16 in s16h is dead by now -- once the down-projection's weight gradient has read it)
            if (int e = side_wait(0)) return e;
            if (int e = launch_ln_bwd_x(l.t1, m->P(b + "/final_layer_norm/gamma"), tmp, dt1, do_tail ? s16h : nullptr, dg2 ? dg2 : t->dummy,
                                        db2 ? db2 : t->dummy + H, BT, H, eps, ln2_ws, s, do_tail ? (gbo ? gbo : t->dummy + 2 * H) : nullptr, dh,
                                        do_tail ? &do_drop : nullptr, fb))
                return e;
            bo_done = do_tail && gbo;
        } else {
            W2V2_HIP_CHECK(hipMemcpyAsync(dt1, dh, (size_t)BT * H * 4, hipMemcpyDeviceToDevice, s));
            do_tail = false;
        }
        float* d_o = tmp;
        if (!do_tail)
            if (int e = dropout_bwd_bias(nullptr, dt1, do16_only ? nullptr : d_o, s16h, BT, H, 0, layer_stream(i, 1), gbo, &bo_done)) return e;
        float* dctx = tmp2;
        // (the bf16 attention backward reads dctx and q | k | v as bf16: the GEMM leaves the dctx shadow, the forward left qkv16)
        uint16_t* const dctx16 = (s16q && dx_shadowed(m->P(b + "/attention/out_proj/kernel"))) ? t->dy16_ctx : nullptr;
        // (c16: the forward kept O only as bf16 -- dctx is then written only as bf16 too and the backward reads both as bf16)
        const bool c16 = t->ctx16_only;
        W2V2_REQUIRE(!c16 || (dctx16 && xs && t->x16_attn), "train_backward: the forward kept the attention output as bf16 only, which needs the shadow paths");
        if (int e = gemm_dx(do16_only ? nullptr : d_o, s16h, H, l.WoT, m->P(b + "/attention/out_proj/kernel"), c16 ? nullptr : dctx, H, nullptr, (int)BT, H, H, s, dctx16)) return e;
        AttnTrain tr{p, seed, layer_stream(i, 0), l.lse, t->x16_attn ? l.keep_bits : nullptr};
        const bool q16 = dqkv16_only(i, l.a16);
        auto out_weight_grad = [&](hipStream_t st, float* rws) {
            return weight_grad(m, t->ctx16_only ? nullptr : l.ctx, do16_only ? nullptr : d_o, (int)BT, H, H, G(b + "/attention/out_proj/kernel"), bo_done ? nullptr : gbo, st,
                               (xs && t->x16_attn) ? l.ctx16 : nullptr, s16h, s16h != nullptr, rws, &ws2);
        };
        // (side stream: the out-projection's weight gradient goes out BEFORE the attention backward, to run under it)
        if (side_on)
            if (int e = side(2, out_weight_grad)) return e;
        if (int e = side_wait(3)) return e;                  // (dqkv / its shadow / the column partials: the layer above's q|k|v weight gradient)
        if (int e = launch_attention_bwd(pf, t->x16_attn ? nullptr : l.qkv, flen, c16 ? nullptr : l.ctx, c16 ? nullptr : dctx, q16 ? nullptr : t->g3h, t->dvec, B, T, H, c.num_heads, tr,
                                         s, s16q, t->x16_attn ? l.qkv16 : nullptr, dctx16, q16 ? t->attn_colpart : nullptr, c16 ? l.ctx16 : nullptr))
            return e;
        // (one stream: the out-projection's weight gradient runs here, not before its data gradient: the attention backward has just read O, so
        //  the GEMM finds it in the Infinity Cache instead of streaming it from HBM cold; d_o / its shadow are untouched until below)
        if (!side_on)
            if (int e = out_weight_grad(s, t->red_ws)) return e;
        if (int e = side(3, [&](hipStream_t st, float* rws) { return qkv_weight_grad(b, l.a, l.a16, q16, st, rws); })) return e;
        if (!do16_only)
            if (int e = side_wait(2)) return e;              // (d_o in fp32 lives in tmp)
        if (int e = gemm_dx(q16 ? nullptr : t->g3h, s16q, 3 * H, l.WqkvT, m->qkv_w[i], tmp, H, nullptr, (int)BT, H, 3 * H, s)) return e;
        float* dg1 = G(b + "/layer_norm/gamma");
        float* db1 = G(b + "/layer_norm/beta");
        // dh = dt1 (residual) + LN1-backward(tmp), one pass (+ the shadow the next iteration's down-projection GEMMs stream)
        dh16_valid = s16h && H % 4 == 0 && (reinterpret_cast<uintptr_t>(dt1) & 15) == 0;
        if (int e = side_wait(2)) return e;                  // (d_o's shadow in s16h: the out-projection's weight gradient)
        if (int e = side_wait(0)) return e;                  // (dh in fp32, when the down-projection's weight gradient read that)
        // (+ the column sums of dh: the down-projection bias gradient of layer i - 1, unless stochastic depth dropped that layer's FFN)
        float* const next_b2 = (shb && tune_int("W2V2_B2_FROM_LN", 1) != 0 && i > 0 && t->layers[i - 1].keep != 0.f)
                                   ? G("encoder/layers/" + std::to_string(i - 1) + "/feed_forward/output_dense/bias") : nullptr;
        if (int e = launch_ln_bwd_x(x, m->P(b + "/layer_norm/gamma"), tmp, dh, dh16_valid ? s16h : nullptr, dg1 ? dg1 : t->dummy,
                                    db1 ? db1 : t->dummy + H, BT, H, eps, ln1_ws, s, next_b2, dt1, nullptr, fb))
            return e;
        b2_done = next_b2 != nullptr;
        if (int e = fold.flush(s)) return e;                 // (the layer's deferred folds: its gradients are final behind this launch)
        if (int e = bucket_done(c.num_layers - i)) return e;
    }

    for (int i = c.num_layers - 1; i >= 0 && !prenorm; --i) {
        const std::string b = "encoder/layers/" + std::to_string(i);
        LayerSave& l = t->layers[i];
        const WgSite ws0 = wg_site(0), ws1 = wg_site(1), ws2 = wg_site(2);
        // hs[i+1] = LN(t3)
        float* dt3 = tmp;
        float* dg2 = G(b + "/final_layer_norm/gamma");
        float* db2 = G(b + "/final_layer_norm/beta");
        // (dt3 is the dY of the FFN down-projection: its column sums are that layer's bias gradient)
        float* const gb2 = (shb && l.keep != 0.f) ? G(b + "/feed_forward/output_dense/bias") : nullptr;
        if (int e = side_wait(2)) return e;                  // (tmp / s16h still hold the d_o the layer above's out-projection weight gradient reads)
        if (int e = launch_ln_bwd_x(l.t3, m->P(b + "/final_layer_norm/gamma"), dh, dt3, (l.keep != 0.f && H % 4 == 0) ? s16h : nullptr,
                                    dg2 ? dg2 : t->dummy, db2 ? db2 : t->dummy + H, BT, H, eps, ln2_ws, s, gb2, nullptr, nullptr, fb))
            return e;
        float* dt2 = tmp2;
        if (l.keep != 0.f) {
            // t3 = t2 + f,  f = gd W2 + b2
            if (int e = side(0, [&](hipStream_t st, float* rws) {
                    return weight_grad(m, f16 ? nullptr : l.gd, dt3, (int)BT, F, H, G(b + "/feed_forward/output_dense/kernel"),
                                       gb2 ? nullptr : G(b + "/feed_forward/output_dense/bias"), st, xs ? l.gd16 : nullptr, H % 4 == 0 ? s16h : nullptr,
                                       s16h != nullptr, rws, &ws0);
                }))
                return e;
            // du = dgd * keep/(1-p) * GELU'(u)   (+ its column sums = the up-projection's bias gradient)
            bool b1_done = false;
            float* const gb1 = G(b + "/feed_forward/intermediate_dense/bias");
            // (f16: du is needed only as bf16 -- both consumers stream the shadow, the column sums come from the producer)
            const bool du16_only = f16 && (!gb1 || shb) && dx_shadowed(m->P(b + "/feed_forward/intermediate_dense/kernel"));
            float* const du = du16_only ? nullptr : t->gf;
            if (int e = side_wait(1)) return e;              // (du / its shadow: read by the layer above's up-projection weight gradient)
            if (int e = ffn_hidden_grad(i, l, b, dt3, H % 4 == 0 ? s16h : nullptr, du16_only, gb1, &b1_done)) return e;
            if (int e = side(1, [&](hipStream_t st, float* rws) {
                    return weight_grad(m, l.t2, du, (int)BT, H, F, G(b + "/feed_forward/intermediate_dense/kernel"),
                                       b1_done ? nullptr : gb1, st, xs ? l.t2_16 : nullptr, s16f, s16f != nullptr, rws, &ws1);
                }))
                return e;
            // dt2 = du W1^T + dt3 (the residual branch)
            if (int e = gemm_dx(du, s16f, F, l.W1T, m->P(b + "/feed_forward/intermediate_dense/kernel"), dt2, H, dt3, (int)BT, H, F, s)) return e;
        } else {
            W2V2_HIP_CHECK(hipMemcpyAsync(dt2, dt3, (size_t)BT * H * 4, hipMemcpyDeviceToDevice, s));
        }
        // t2 = LN(t1)
        float* dt1 = tmp3;
        float* dg1 = G(b + "/layer_norm/gamma");
        float* db1 = G(b + "/layer_norm/beta");
        // t1 = dropout(o) + x,  o = ctx Wo + bo
        float* d_o = tmp;     // dt3 (and its shadow) is dead
        bool bo_done = false;
        float* const gbo = G(b + "/attention/out_proj/bias");
        // (d_o is read only as bf16 when both of its GEMMs stream shadows and its column sums come from the producer: the
        //  dropout backward that makes it then rides in the tail of the LayerNorm backward that produces dt1)
        const bool do16_only = xs && t->x16_attn && s16h && H % 128 == 0 && (BT * H) % 4 == 0 && dx_shadowed(m->P(b + "/attention/out_proj/kernel"));
        const bool do_tail = do16_only && shb && fuse_do_tail;
        const LnDropTail do_drop{p, seed, layer_stream(i, 1)};
        if (int e = side_wait(0)) return e;                  // (dt3 in tmp / s16h: the down-projection's weight gradient; d_o goes there next)
        if (int e = launch_ln_bwd_x(l.t1, m->P(b + "/layer_norm/gamma"), dt2, dt1, do_tail ? s16h : nullptr, dg1 ? dg1 : t->dummy,
                                    db1 ? db1 : t->dummy + H, BT, H, eps, ln1_ws, s, do_tail ? (gbo ? gbo : t->dummy + 2 * H) : nullptr, nullptr,
                                    do_tail ? &do_drop : nullptr, fb))
            return e;
        bo_done = do_tail && gbo;
        if (!do_tail)
            if (int e = dropout_bwd_bias(nullptr, dt1, do16_only ? nullptr : d_o, s16h, BT, H, 0, layer_stream(i, 1), gbo, &bo_done)) return e;
        float* dctx = tmp2;   // dt2 is dead
        // (the bf16 attention backward reads dctx and q | k | v as bf16: the GEMM leaves the dctx shadow, the forward left qkv16)
        uint16_t* const dctx16 = (s16q && dx_shadowed(m->P(b + "/attention/out_proj/kernel"))) ? t->dy16_ctx : nullptr;
        const bool c16 = t->ctx16_only;
        W2V2_REQUIRE(!c16 || (dctx16 && xs && t->x16_attn), "train_backward: the forward kept the attention output as bf16 only, which needs the shadow paths");
        if (int e = gemm_dx(do16_only ? nullptr : d_o, s16h, H, l.WoT, m->P(b + "/attention/out_proj/kernel"), c16 ? nullptr : dctx, H, nullptr, (int)BT, H, H, s, dctx16)) return e;
        AttnTrain tr{p, seed, layer_stream(i, 0), l.lse, t->x16_attn ? l.keep_bits : nullptr};
        const uint16_t* const hs16_i = m->hs16.size() > (size_t)i ? m->hs16[i] : nullptr;
        const bool q16 = dqkv16_only(i, hs16_i);
        auto out_weight_grad = [&](hipStream_t st, float* rws) {
            return weight_grad(m, t->ctx16_only ? nullptr : l.ctx, do16_only ? nullptr : d_o, (int)BT, H, H, G(b + "/attention/out_proj/kernel"), bo_done ? nullptr : gbo, st,
                               (xs && t->x16_attn) ? l.ctx16 : nullptr, s16h, s16h != nullptr, rws, &ws2);
        };
        // (side stream: the out-projection's weight gradient goes out BEFORE the attention backward, to run under it)
        if (side_on)
            if (int e = side(2, out_weight_grad)) return e;
        if (int e = side_wait(3)) return e;                  // (dqkv / its shadow / the column partials: the layer above's q|k|v weight gradient)
        if (int e = launch_attention_bwd(pf, t->x16_attn ? nullptr : l.qkv, flen, c16 ? nullptr : l.ctx, c16 ? nullptr : dctx, q16 ? nullptr : t->g3h, t->dvec, B, T, H, c.num_heads, tr,
                                         s, s16q, t->x16_attn ? l.qkv16 : nullptr, dctx16, q16 ? t->attn_colpart : nullptr, c16 ? l.ctx16 : nullptr))
            return e;
        if (!side_on)
            if (int e = out_weight_grad(s, t->red_ws)) return e;
        if (int e = side(3, [&](hipStream_t st, float* rws) { return qkv_weight_grad(b, m->hs[i], hs16_i, q16, st, rws); })) return e;
        // dx = dqkv Wqkv^T + dt1 (residual)
        if (int e = gemm_dx(q16 ? nullptr : t->g3h, s16q, 3 * H, l.WqkvT, m->qkv_w[i], dh, H, dt1, (int)BT, H, 3 * H, s)) return e;
        if (int e = fold.flush(s)) return e;                 // (the layer's deferred folds: its gradients are final behind this launch)
        if (int e = bucket_done(c.num_layers - i)) return e;
    }
    // (the scratch the layers' gradients flowed through is reused from here on, and the last weight gradient below shares the slabs)
    for (int k = 0; k < 4; ++k)
        if (int e = side_wait(k)) return e;
    // ---- encoder input: postnorm hs[0] = dropout(LN(posout));  prenorm hs0 = dropout(posout) ----
    float* dpos = tmp2;
    if (prenorm) {
        if (int e = launch_dropout_bwd(nullptr, dh, dpos, BT * H, 0, p, seed, DS_ENCODER_IN, s)) return e;
    } else {
        if (int e = launch_dropout_bwd(nullptr, dh, tmp, BT * H, 0, p, seed, DS_ENCODER_IN, s)) return e;
        float* dg = G("encoder/layer_norm/gamma");
        float* db = G("encoder/layer_norm/beta");
        if (int e = launch_ln_bwd(m->posout, m->P("encoder/layer_norm/gamma"), tmp, dpos, dg ? dg : t->dummy, db ? db : t->dummy + H, BT, H,
                                  eps, t->red_ws, s))
            return e;
    }
    // ---- positional conv: posout = xz + GELU(c),  c = conv(xz; W_eff) + bias ----
    const int K = c.num_conv_pos_embeddings, Gr = c.num_conv_pos_embedding_groups, cg = H / Gr;
    float* dc = tmp;
    if (int e = launch_dropout_bwd(t->pos_c, dpos, dc, BT * H, act_ew, 0.f, 0, 0, s)) return e;       // dc = dpos * GELU'(c)
    if (float* gb = G("encoder/pos_conv_embed/conv/bias"))
        if (int e = launch_colsum(dc, gb, BT, H, t->red_ws, 0, s)) return e;
    const float* enc_x = t->have_spec ? t->hm : t->hd;
    const float* xz = enc_x;
    if (flen) {
        if (int e = launch_mask_rows(enc_x, flen, tmp3, B, T, H, s)) return e;
        xz = tmp3;
    }
    float* gv = G("encoder/pos_conv_embed/conv/weight_v");
    float* gg = G("encoder/pos_conv_embed/conv/weight_g");
    if (gv || gg) {
        if (w2v2_pos_conv_bf16_ok(m) && B <= 64) {
            // precision mode 1: batched transposed-A GEMM on the bf16 pipe (one (K cg, og) slab per sample and group, summed after);
            // the frames are padded to a multiple of 64 with zero rows of dc (T = 1499 at 480000 samples)
            const int Tk = (T + 63) / 64 * 64;
            if (!t->pos_pack32) {
                if (int e = t_alloc(t, &t->pos_pack32, (int64_t)B * (Tk + K - 1) * H)) return e;
                // (one (K cg, og) slab per group and per SLAB of samples: at most four since the samples are concatenated along K -- posconv.hip;
                //  the one-slab-per-sample form of the tools-only build keeps B of them)
                const int nslab = tune_int("W2V2_POS_DW_KCAT", 1) != 0 ? (B < 4 ? B : 4) : B;
                if (int e = t_alloc(t, &t->pos_dw_slabs, (int64_t)nslab * K * cg * H)) return e;
                if (Tk != T)
                    if (int e = t_alloc(t, &t->pos_dc_pad, (int64_t)B * Tk * H)) return e;
            }
            if (int e = launch_pos_conv_dw_bf16(pf, xz, dc, t->dwg, t->pos_pack32, t->pos_dw_slabs, t->red_ws, B, T, H, K, Gr, s, t->pos_dc_pad))
                return e;
        } else if (int e = launch_pos_conv_dw(pf, xz, dc, t->dwg, nullptr, B, T, H, K, Gr, s)) {
            return e;
        }
        float* gvd = gv ? gv : t->dwv_scratch;   // scratch targets when only one of the pair trains
        float* ggd = gg ? gg : t->dummy;
        if (int e = launch_weight_norm_bwd(m->P("encoder/pos_conv_embed/conv/weight_v"), m->P("encoder/pos_conv_embed/conv/weight_g"),
                                           t->dwg, gvd, ggd, K, cg, H, Gr, s))
            return e;
    }
    // dxz = dpos + conv^T(dc)   (transposed kernel, pad_left = K - 1 - K/2), then the frame mask
    float* dxz = dh;
    if (w2v2_pos_conv_bf16_ok(m) && m->pos_pack16) {
        if (!t->pos_w16_t) W2V2_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&t->pos_w16_t), (size_t)K * cg * H * sizeof(uint16_t)));
        if (!t->pos_w16_t_fresh) {
            if (int e = launch_pos_conv_weight_shadow(t->pos_wg_t, t->pos_w16_t, K, cg, Gr, s)) return e;
            t->pos_w16_t_fresh = true;
        }
        if (int e = launch_pos_conv_bf16(pf, dc, t->pos_w16_t, nullptr, nullptr, dxz, nullptr, m->pos_pack16, nullptr, B, T, H, K, Gr, 0,
                                         K - 1 - K / 2, 0, s))
            return e;
    } else if (int e = launch_pos_conv_ex(pf, dc, t->pos_wg_t, nullptr, nullptr, dxz, nullptr, B, T, H, K, Gr, 0, K - 1 - K / 2, 0, s)) {
        return e;
    }
    if (int e = launch_axpby(dxz, dpos, dxz, BT * H, 1.f, 1.f, s)) return e;
    if (flen)
        if (int e = launch_mask_rows(dxz, flen, dxz, B, T, H, s)) return e;
    // ---- spec-augment ----
    float* dhd = dxz;
    if (t->have_spec) {
        if (int e = launch_spec_aug_bwd(dxz, t->spec_mask, tmp, tmp2, BT, H, s)) return e;
        if (float* ge = G("masked_spec_embed"))
            if (int e = launch_colsum(tmp2, ge, BT, H, t->red_ws, 0, s)) return e;
        dhd = tmp;
    }
    // ---- feature projection: hd = dropout(ln512 Wp + bp),  ln512 = LN(conv_out) ----
    float* dproj = tmp3;
    if (int e = launch_dropout_bwd(nullptr, dhd, dproj, BT * H, 0, p, seed, DS_FEATURE_PROJECTION, s)) return e;
    const int C = c.filter_sizes[c.num_conv_layers - 1];
    if (int e = weight_grad(m, m->ln512, dproj, (int)BT, C, H, G("feature_projection/projection/kernel"),
                            G("feature_projection/projection/bias"), s))
        return e;
    float* dgp = G("feature_projection/layer_norm/gamma");
    float* dbp = G("feature_projection/layer_norm/beta");
    if (dgp || dbp) {
        float* dln = tmp2;       // (BT, C) fits a (BT, H) buffer when C <= H; otherwise use the FFN scratch
        if (C > H) dln = t->gf;
        if (int e = gemm_dx(dproj, nullptr, H, t->WpT, m->P("feature_projection/projection/kernel"), dln, C, nullptr, (int)BT, C, H, s)) return e;
        float* dxc = C > H ? t->g3h : tmp;    // gradient w.r.t. the frozen conv output: computed and dropped
        if (int e = launch_ln_bwd(m->conv[c.num_conv_layers - 1], m->P("feature_projection/layer_norm/gamma"), dln, dxc,
                                  dgp ? dgp : t->dummy, dbp ? dbp : t->dummy + C, BT, C, eps, t->red_ws, s))
            return e;
    }
    return bucket_done(nbuckets - 1);
}

/* Gradient buckets: slices of the flat buffer in the order the backward completes them.  Data-parallel callers enqueue
 * one all-reduce per bucket on a communication stream that waits (w2v2_train_bucket_wait) for that bucket only, so the
 * collectives of the upper layers run under the backward of the lower ones. */
int w2v2_train_storage(const w2v2_model* m, int32_t* mask, int64_t* workspace_bytes) {
    W2V2_REQUIRE(m && mask && workspace_bytes, "train_storage: null argument");
    const TrainState* t = m->train;
    *mask = 0;
    *workspace_bytes = t ? t->alloc_bytes : 0;
    if (t && t->B > 0)
        *mask = (t->x16_attn ? W2V2_TRAIN_BF16_QKV : 0) | (t->ctx16_only ? W2V2_TRAIN_BF16_CTX : 0) | (t->ffn16_only ? W2V2_TRAIN_BF16_FFN : 0) |
                (t->u16_only ? W2V2_TRAIN_BF16_U : 0) | (t->ln16_only ? W2V2_TRAIN_BF16_LN : 0);
    return W2V2_OK;
}

int w2v2_train_num_buckets(const w2v2_model* m) { return m ? m->cfg.num_layers + 2 : W2V2_EINVAL; }

int w2v2_train_bucket(w2v2_model* m, int32_t k, int64_t* offset, int64_t* numel) {
    W2V2_REQUIRE(m && offset && numel, "train_bucket: null argument");
    TrainState* t = get_state(m);
    const int nl = m->cfg.num_layers, nb = nl + 2;
    W2V2_REQUIRE(k >= 0 && k < nb, "train_bucket: bucket %d outside [0, %d)", k, nb);
    // inventory order: [front: everything before encoder/layers/0][layers 0 .. N-1][lm_head]
    auto first_with = [&](const std::string& prefix) -> int64_t {
        for (size_t i = 0; i < m->params.size(); ++i)
            if (m->params[i].name.compare(0, prefix.size(), prefix) == 0) return t->goff[i];
        return t->gtotal;
    };
    auto layer_begin = [&](int i) { return i < nl ? first_with("encoder/layers/" + std::to_string(i) + "/") : first_with("lm_head/"); };
    int64_t lo, hi;
    if (k == 0) { lo = first_with("lm_head/"); hi = t->gtotal; }
    else if (k <= nl) { const int i = nl - k; lo = layer_begin(i); hi = layer_begin(i + 1); }
    else { lo = 0; hi = layer_begin(0); }
    W2V2_REQUIRE(lo <= hi, "train_bucket: the variable inventory is not in [front | layers | lm_head] order");
    *offset = lo;
    *numel = hi - lo;
    return W2V2_OK;
}

/* Slot of one variable in the flat gradient / Adam buffers (16-byte aligned slots, inventory order). */
int w2v2_grad_slot(w2v2_model* m, const char* name, int64_t* offset, int64_t* numel) {
    W2V2_REQUIRE(m && name && offset && numel, "grad_slot: null argument");
    TrainState* t = get_state(m);
    auto it = m->index.find(name);
    if (it == m->index.end()) {
        set_error("grad_slot: unknown variable `%s`", name);
        return W2V2_ENOTFOUND;
    }
    *offset = t->goff[it->second];
    *numel = m->params[it->second].numel;
    return W2V2_OK;
}

int w2v2_train_bucket_wait(w2v2_model* m, int32_t k, void* stream) {
    W2V2_REQUIRE(m && m->train, "train_bucket_wait: no training state");
    TrainState* t = m->train;
    W2V2_REQUIRE(k >= 0 && k < (int)t->bucket_ev.size(), "train_bucket_wait: bucket %d has not been produced yet (run a backward first)", k);
    W2V2_HIP_CHECK(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), t->bucket_ev[k], 0));
    return W2V2_OK;
}

/* A fresh optimizer: the reference builds a new tf.keras.optimizers.Adam for each stage (src/main.py:213,240), i.e. zero
 * moments and iteration 0.  The moments live here (not in the Python Trainer), so a new Trainer on a used model resets them. */
int w2v2_adam_reset(w2v2_model* m, void* stream) {
    W2V2_REQUIRE(m, "adam_reset: null model");
    if (int e = ensure_persistent(m)) return e;
    TrainState* t = m->train;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    W2V2_HIP_CHECK(hipMemsetAsync(t->adam_m, 0, (size_t)t->gtotal * 4, s));
    W2V2_HIP_CHECK(hipMemsetAsync(t->adam_v, 0, (size_t)t->gtotal * 4, s));
    return W2V2_OK;
}

/* Adam's first / second moment buffers (flat, the gradient buffer's layout) for checkpoint / resume. */
int w2v2_adam_buffers(w2v2_model* m, float** m_dev, float** v_dev, int64_t* numel) {
    W2V2_REQUIRE(m && m_dev && v_dev && numel, "adam_buffers: null argument");
    if (int e = ensure_persistent(m)) return e;        // the moments do not depend on the batch shape: no forward needed
    TrainState* t = m->train;
    *m_dev = t->adam_m;
    *v_dev = t->adam_v;
    *numel = t->gtotal;
    return W2V2_OK;
}

int w2v2_grad_buffer(w2v2_model* m, float** dev_ptr, int64_t* numel) {
    W2V2_REQUIRE(m && dev_ptr && numel, "grad_buffer: null argument");
    if (int e = ensure_persistent(m)) return e;
    TrainState* t = m->train;
    *dev_ptr = t->grads;
    *numel = t->gtotal;
    return W2V2_OK;
}

int w2v2_get_grad(w2v2_model* m, const char* name, float* host_dst, int64_t numel, void* stream) {
    W2V2_REQUIRE(m && name && host_dst, "get_grad: null argument");
    TrainState* t = m->train;
    if (!t || !t->grads) {
        set_error("get_grad: run a training forward/backward first");
        return W2V2_ESTATE;
    }
    auto it = m->index.find(name);
    if (it == m->index.end()) {
        set_error("get_grad: unknown variable `%s`", name);
        return W2V2_ENOTFOUND;
    }
    W2V2_REQUIRE(numel == m->params[it->second].numel, "get_grad: element count mismatch for `%s`", name);
    W2V2_HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    W2V2_HIP_CHECK(hipMemcpy(host_dst, t->grads + t->goff[it->second], (size_t)numel * 4, hipMemcpyDeviceToHost));
    return W2V2_OK;
}

int w2v2_adam_step(w2v2_model* m, float lr, float beta1, float beta2, float eps, int64_t step, void* stream) {
    W2V2_REQUIRE(m && step >= 1, "adam_step: bad argument");
    TrainState* t = m->train;
    if (!t || !t->grads) {
        set_error("adam_step: no gradients");
        return W2V2_ESTATE;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    StepProfScope step_prof(m->prof);
    // Keras: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  p -= lr_t * m / (sqrt(v) + eps)
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    if (!t->adam_table_fresh) {
        std::vector<AdamChunk> host;
        for (size_t i = 0; i < m->params.size(); ++i) {
            if (!t->trainable[i]) continue;
            const Param& p = m->params[i];
            for (int64_t o = 0; o < p.numel; o += 4096)
                host.push_back(AdamChunk{p.dev + o, t->goff[i] + o, (int)(p.numel - o < 4096 ? p.numel - o : 4096)});
        }
        if (t->adam_chunks) (void)hipFree(t->adam_chunks);
        t->adam_chunks = nullptr;
        t->adam_nchunks = (int)host.size();
        if (!host.empty()) {
            W2V2_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&t->adam_chunks), host.size() * sizeof(AdamChunk)));
            W2V2_HIP_CHECK(hipMemcpy(t->adam_chunks, host.data(), host.size() * sizeof(AdamChunk), hipMemcpyHostToDevice));
        }
        t->adam_table_fresh = true;
    }
    if (t->adam_nchunks > 0)
        if (int e = launch_adam_multi(t->adam_chunks, t->adam_nchunks, t->grads, t->adam_m, t->adam_v, (float)lr_t, beta1, beta2, eps, s))
            return e;
    t->transposes_fresh = false;
    m->finalized = false;
    return w2v2_finalize(m, stream);     // re-derive the effective positional kernel and the packed q|k|v
}

int64_t w2v2_ln_bwd_ws_floats(int64_t rows, int32_t C) { return ln_bwd_ws_floats(rows, C); }
int w2v2_op_layer_norm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma,
                           float* dbeta, int64_t rows, int32_t C, float eps, float* ws, void* stream) {
    return launch_ln_bwd(x, gamma, dy, dx, dgamma, dbeta, rows, C, eps, ws, reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_attention_train(const float* qkv, const int32_t* frame_len, float* ctx, float* lse, int32_t B, int32_t T,
                            int32_t H, int32_t heads, float p, uint64_t seed, uint32_t stream_id, void* stream) {
    AttnTrain tr{p, seed, stream_id, lse};
    return launch_attention_train(nullptr, qkv, frame_len, ctx, B, T, H, heads, tr, reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_attention_bwd(const float* qkv, const int32_t* frame_len, const float* ctx, const float* lse,
                          const float* dctx, float* dqkv, float* dvec_ws, int32_t B, int32_t T, int32_t H,
                          int32_t heads, float p, uint64_t seed, uint32_t stream_id, void* stream) {
    AttnTrain tr{p, seed, stream_id, const_cast<float*>(lse)};
    return launch_attention_bwd(nullptr, qkv, frame_len, ctx, dctx, dqkv, dvec_ws, B, T, H, heads, tr,
                                reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_layer_norm_dropout(const float* x, const float* residual, float* t1, float* y, uint16_t* y16, const float* gamma, const float* beta,
                               int64_t rows, int32_t C, float eps, float p, uint64_t seed, uint32_t stream_id, void* stream) {
    return launch_layer_norm_drop(nullptr, x, residual, t1, y, y16, gamma, beta, rows, C, eps, p, seed, stream_id, reinterpret_cast<hipStream_t>(stream));
}
int w2v2_op_dropout(const float* x, const float* residual, float* y, int64_t n, int32_t act, float p,
                    uint64_t seed, uint32_t stream_id, void* stream) {
    return launch_dropout_fwd(x, residual, y, n, act, p, seed, stream_id, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"

// What a data-parallel SUM all-reduce of bucket k has to send (comm.hip; the host-side mirror is wav2vec2/dist.py::trainable_ranges):
// the runs of trainable slots, adjacent slots merged (a slot is its variable rounded up to 4 floats; the padding rides along, it is
// zero on every rank).  Frozen slots -- the conv stack in stage 2 (main.py:234-237), everything but lm_head in stage 1 -- stay home.
int w2v2_train_trainable_runs(w2v2_model* m, int k, std::vector<std::pair<int64_t, int64_t>>* runs, float** grads) {
    W2V2_REQUIRE(m && runs, "trainable_runs: null argument");
    int64_t lo = 0, n = 0;
    if (int e = w2v2_train_bucket(m, k, &lo, &n)) return e;
    if (int e = ensure_persistent(m)) return e;
    TrainState* t = m->train;
    const int64_t hi = lo + n;
    runs->clear();
    for (size_t i = 0; i < m->params.size(); ++i) {
        const int64_t off = t->goff[i];
        if (off < lo || off >= hi || !t->trainable[i]) continue;
        int64_t end = off + ((m->params[i].numel + 3) & ~(int64_t)3);
        if (end > hi) end = hi;
        if (!runs->empty() && runs->back().first + runs->back().second == off) runs->back().second = end - runs->back().first;
        else runs->emplace_back(off, end - off);
    }
    if (grads) *grads = t->grads;
    return W2V2_OK;
}

