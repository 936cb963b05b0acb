// Declarations for the training-step kernels (train_kernels.hip, attention_bwd in attention.hip,
// posconv.hip) and the shared counter-based dropout hash.
#pragma once

#include "common.h"

namespace w2v2 {

// Dropout sites ("streams") of the training forward; the hash key is (seed, stream, element index).
enum DropStream : uint32_t {
    DS_FEATURE_PROJECTION = 1,   // feature_extractor.py:95
    DS_ENCODER_IN = 2,           // encoder.py:270
    DS_HEAD = 3,                 // modeling.py:253
    DS_LAYER_BASE = 16,          // + 4 * layer + {0: attention probs, 1: attention output, 2: FFN intermediate}
};

#ifdef __HIPCC__
// keep(seed, stream, idx, p): counter-based hash, 32-bit arithmetic only.  History of its cost in the one place that is bound by it (the
// bf16 attention forward evaluates it once per score): 64-bit splitmix per element, ~40 VALU ops; one lowbias32 per element, ~16; one
// per PAIR (round 3); round 6: one mixer ROUND per OCT of eight consecutive elements, finished by two 32 x 32 -> 64-bit products that
// yield four words = eight 16-bit decisions (2.25 plain ops per decision against 4.8: VERDICT r05 item 2):
//   k        = low 32 bits of splitmix64(seed ^ stream * GOLD)                          -- loop-invariant
//   x        = round1(lo32(idx >> 3) * 0x9E3779B1 ^ k),  round1(x): x ^= x >> 16; x *= 0x7feb352d; x ^= x >> 15   (lowbias32's first half)
//   (lo, hi) = x * 0x846ca68b  (words 0, 1)   |   x * 0xC2B2AE35  (words 2, 3)           -- the full 64-bit products
//   word 2j  = lo ^ (lo >> 16)               (= lowbias32's second half for j = 0)
//   word 2j+1 = hi + (lo << 16)              (hi alone is not uniform -- it is < the multiplier; its sum with the product's low half is)
//   keep     = ((idx odd ? w >> 16 : w & 0xFFFF) ^ 0x8000) >= floor(p * 2^16),  w = word (idx >> 1) & 3 of oct idx >> 3
// p is realised to 2^-16 (0.1 -> 0.1000061).  Measured on 6-12 M decisions per key (six keys, tools/dropout_hash_stats.py: the tests the
// pair hash passed): keep-rate error < 2e-4, lag / column / diagonal / in-oct pair correlations at the sampling noise (< 1e-3; < 4e-3 for
// the maximum over the 28 in-oct pairs), 16-bit halves uniform (chi-square, 255 dof: 280-300).  The index enters modulo 2^32 (a pattern
// repeats after 4.3e9 elements of one tensor).  Attention probabilities index their (B heads T, T) matrix with the row stride rounded up
// to a multiple of 16 and bits 2 and 3 of the key index exchanged (attention_drop_stride / attention_drop_col): an oct is then the eight
// scores a lane of the forward holds in accumulator registers 8a .. 8a + 7.  Same integer function as
// wav2vec2/variables.py::dropout_hash / attention_keep.
constexpr uint32_t DROPOUT_FIB = 0x9E3779B1u;
constexpr uint32_t DROPOUT_M1 = 0x7feb352du, DROPOUT_M2 = 0x846ca68bu, DROPOUT_M3 = 0xC2B2AE35u;
__device__ __forceinline__ uint32_t dropout_key(uint64_t seed, uint32_t stream) {
    uint64_t z = (seed ^ ((uint64_t)stream * 0x9E3779B97F4A7C15ULL)) + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return (uint32_t)(z ^ (z >> 31));
}
__device__ __forceinline__ uint32_t dropout_threshold(float p) {      // 16-bit threshold
    return (uint32_t)((double)p * 65536.0);
}
// the shared round of an oct, from the pre-multiplied oct index (oct * 0x9E3779B1 mod 2^32: callers whose oct indices are `base + small
// constant` pay the multiply once per base and an add per oct -- products distribute over the sum modulo 2^32)
__device__ __forceinline__ uint32_t dropout_oct_x(uint32_t key, uint32_t oct_times_fib) {
    uint32_t x = oct_times_fib ^ key;
    x ^= x >> 16; x *= DROPOUT_M1; x ^= x >> 15;
    return x;
}
// words 2 H and 2 H + 1 of the oct (elements 4 H .. 4 H + 3): one 64-bit product
template <int H>
__device__ __forceinline__ void dropout_oct_words(uint32_t x, uint32_t& w_even, uint32_t& w_odd) {
    const uint64_t prod = (uint64_t)x * (uint64_t)(H ? DROPOUT_M3 : DROPOUT_M2);
    const uint32_t lo = (uint32_t)prod, hi = (uint32_t)(prod >> 32);
    w_even = lo ^ (lo >> 16);
    w_odd = hi + (lo << 16);
}
// the hash word of element pair (2 pair, 2 pair + 1): low half decides the even element, high half the odd one
__device__ __forceinline__ uint32_t dropout_word(uint32_t key, uint32_t pair) {
    const uint32_t x = dropout_oct_x(key, (pair >> 2) * DROPOUT_FIB);
    const uint64_t prod = (uint64_t)x * (uint64_t)((pair & 2u) ? DROPOUT_M3 : DROPOUT_M2);
    const uint32_t lo = (uint32_t)prod, hi = (uint32_t)(prod >> 32);
    return (pair & 1u) ? hi + (lo << 16) : lo ^ (lo >> 16);
}
// The decision reads a 16-bit half as a SIGNED number: keep  <=>  int16(half) >= thr - 2^15  <=>  (half ^ 0x8000) >= thr  (round 4;
// before: half >= thr -- the same probability, the top bit of the uniform half flipped).  In this form the decisions of BOTH halves of a
// word come out of two packed 16-bit instructions (dropout_keep_mask_pk below), which is what the bf16 attention forward -- VALU-bound,
// 60 % of it this hash and its bookkeeping -- needs.  Same function on the host: wav2vec2/variables.py::dropout_hash.
__device__ __forceinline__ bool dropout_keep_lo(uint32_t w, uint32_t thr) { return ((w & 0xFFFFu) ^ 0x8000u) >= thr; }
__device__ __forceinline__ bool dropout_keep_hi(uint32_t w, uint32_t thr) { return ((w >> 16) ^ 0x8000u) >= thr; }
// general form (one hash per call): key and threshold hoisted by the caller, 32-bit index
__device__ __forceinline__ bool dropout_keep32(uint32_t key, uint32_t idx, uint32_t thr) {
    const uint32_t w = dropout_word(key, idx >> 1);
    return (((idx & 1u) ? (w >> 16) : (w & 0xFFFFu)) ^ 0x8000u) >= thr;
}
// Both decisions of a word at once, as a mask: 0xFFFF in the half whose element is KEPT, 0 in the other.  thr1s_pk = dropout_thr1s_pk(thr)
// holds int16(thr - 2^15 - 1) in both halves (thr >= 1): keep  <=>  thr - 2^15 - 1 - int16(half) < 0, a saturating packed subtract
// (v_pk_sub_i16 clamp: the difference overflows 16 bits) and a packed arithmetic shift (v_pk_ashrrev_i16).  The mask has the layout
// of a packed bf16 pair (even element low, odd element high): P pairs are masked AFTER v_cvt_pk_bf16_f32 with one AND.
typedef short w2v2_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dropout_thr1s_pk(uint32_t thr) {
    const uint32_t t = (thr - 32769u) & 0xFFFFu;
    return t | (t << 16);
}
__device__ __forceinline__ uint32_t dropout_keep_mask_pk(uint32_t w, uint32_t thr1s_pk) {
    w2v2_short2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(w2v2_short2, thr1s_pk), __builtin_bit_cast(w2v2_short2, w));
    d = d >> 15;
    return __builtin_bit_cast(uint32_t, d);
}
// index space of the attention probabilities: row stride = T rounded up to a multiple of 16, and inside a row key k sits at column
// attention_drop_col(k) = k with bits 2 and 3 exchanged (see the header of this section: a lane of the bf16 forward holds keys
// 16 a + 8 b + 4 lh + c in register 8 a + 4 b + c, so its registers 8 a .. 8 a + 7 are the consecutive columns 16 a + 8 lh + 0 .. 7 = one oct)
__device__ __forceinline__ uint32_t attention_drop_stride(int T) { return (uint32_t)((T + 15) & ~15); }
__device__ __forceinline__ uint32_t attention_drop_col(uint32_t k) { return (k & ~12u) | ((k & 4u) << 1) | ((k & 8u) >> 1); }
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint32_t stream, uint64_t idx, float p) {
    return dropout_keep32(dropout_key(seed, stream), (uint32_t)idx, dropout_threshold(p));
}

// derivative of the activations of apply_act (common.h); shared by the element-wise kernels and the GEMM training epilogue
__device__ __forceinline__ float gelu_grad(float u, int act) {
    if (act == 1) {   // d/du [0.5 u (1 + erf(u / sqrt 2))]
        const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752440f));
        return cdf + u * 0.39894228040143267794f * expf(-0.5f * u * u);
    }
    if (act == 2) {
        const float c = 0.79788456080286535588f, k = 0.044715f;
        const float t = tanhf(c * (u + k * u * u * u));
        return 0.5f * (1.0f + t) + 0.5f * u * (1.0f - t * t) * c * (1.0f + 3.0f * k * u * u);
    }
    if (act == 3) {   // act 1 evaluated the fast way (precision mode 1): erf by Abramowitz-Stegun 7.1.26, one exponential shared with the density
        const float z = fabsf(u) * 0.70710678118654752440f;
        const float tt = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
        float q = fmaf(1.061405429f, tt, -1.453152027f);
        q = fmaf(q, tt, 1.421413741f);
        q = fmaf(q, tt, -0.284496736f);
        q = fmaf(q, tt, 0.254829592f);
        const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);      // exp(-u^2 / 2)
        const float erf_abs = fmaf(-q * tt, e, 1.0f);
        const float cdf = 0.5f + copysignf(0.5f * erf_abs, u);
        return fmaf(u * 0.39894228040143267794f, e, cdf);
    }
    return 1.0f;
}

// gelu_grad(., 3) on two values at once (packed fp32 multiply / fma; rcp and exp2 stay scalar)
__device__ __forceinline__ f32x2_t gelu_grad_fast2(f32x2_t u) {
    const f32x2_t au = {fabsf(u[0]), fabsf(u[1])};
    const f32x2_t z = au * f32x2_t{0.70710678118654752440f, 0.70710678118654752440f};
    const f32x2_t d = __builtin_elementwise_fma(f32x2_t{0.3275911f, 0.3275911f}, z, f32x2_t{1.0f, 1.0f});
    const f32x2_t t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2_t q = __builtin_elementwise_fma(f32x2_t{1.061405429f, 1.061405429f}, t, f32x2_t{-1.453152027f, -1.453152027f});
    q = __builtin_elementwise_fma(q, t, f32x2_t{1.421413741f, 1.421413741f});
    q = __builtin_elementwise_fma(q, t, f32x2_t{-0.284496736f, -0.284496736f});
    q = __builtin_elementwise_fma(q, t, f32x2_t{0.254829592f, 0.254829592f});
    const f32x2_t zz = z * z * f32x2_t{-1.44269504088896340736f, -1.44269504088896340736f};
    const f32x2_t e = {__builtin_amdgcn_exp2f(zz[0]), __builtin_amdgcn_exp2f(zz[1])};        // exp(-u^2 / 2)
    const f32x2_t herf = __builtin_elementwise_fma(-(q * t), e, f32x2_t{1.0f, 1.0f}) * f32x2_t{0.5f, 0.5f};
    const f32x2_t cdf = {0.5f + copysignf(herf[0], u[0]), 0.5f + copysignf(herf[1], u[1])};
    return __builtin_elementwise_fma(u * f32x2_t{0.39894228040143267794f, 0.39894228040143267794f}, e, cdf);
}
#endif

// one 4096-element piece of one variable for the single-launch Adam: p = variable + offset, goff = its offset in the
// flat gradient / moment buffers
struct AdamChunk {
    float* p;
    int64_t goff;
    int n;
};
int launch_adam_multi(const AdamChunk* chunks_dev, int nchunks, const float* grads, float* m, float* v, float lr_t, float b1,
                      float b2, float eps, hipStream_t s);
// bf16-stored inputs of the element-wise dropout kernels (precision mode 1 keeps the FFN pre-activation u and the gradient of the
// FFN hidden activation only as bf16): a16 stands in for x (forward) / u (backward), b16 for dy (backward; it may be the dx16 the
// call writes -- in place); round_in = the fp32 inputs are rounded to bf16 on the way in (the shadow-free path of the same mode,
// which must produce the same bits).
struct EwBf16 {
    const uint16_t* a16 = nullptr;
    const uint16_t* b16 = nullptr;
    int round_in = 0;
};
int launch_dropout_fwd_x(const float* x, const float* res, float* y, uint16_t* y16 /* optional bf16 shadow */, int64_t n, int act,
                         float p, uint64_t seed, uint32_t stream_id, hipStream_t s, const EwBf16& in = EwBf16{});
int launch_dropout_fwd(const float* x, const float* res, float* y, int64_t n, int act, float p,
                       uint64_t seed, uint32_t stream_id, hipStream_t s);
int launch_dropout_bwd(const float* u, const float* dy, float* dx, int64_t n, int act, float p,
                       uint64_t seed, uint32_t stream_id, hipStream_t s);
int launch_dropout_bwd_x(const float* u, const float* dy, float* dx, uint16_t* dx16 /* optional bf16 shadow */, int64_t n, int act,
                         float p, uint64_t seed, uint32_t stream_id, hipStream_t s, const EwBf16& in = EwBf16{});
int launch_dropout_bwd_colsum(const float* u, const float* dy, float* dx, uint16_t* dx16, float* colsum, int64_t rows, int cols,
                              int act, float p, uint64_t seed, uint32_t stream_id, float* ws, hipStream_t s, const EwBf16& in = EwBf16{},
                              struct FoldBatch* defer = nullptr /* the fold of ws joins this batch (ws then stays live until its flush) */);
// t1 = dropout(a) + res and y (/ y16) = LayerNorm(t1) in one pass (layernorm.hip); bit-identical to launch_dropout_fwd + launch_layer_norm_x
int launch_layer_norm_drop(Profiler* prof, const float* a, const float* res, float* t1, float* y, uint16_t* y16 /* optional bf16 shadow */,
                           const float* gamma, const float* beta, int64_t rows, int C, float eps, float p, uint64_t seed, uint32_t stream,
                           hipStream_t s);
int launch_transpose(const float* x, float* y, int rows, int cols, int nbatch, hipStream_t s);
int64_t colsum_ws_floats(int64_t rows, int cols);
int64_t dropout_bwd_colsum_ws_floats(int64_t rows, int cols);      // scratch of launch_dropout_bwd_colsum
int launch_colsum(const float* x, float* out, int64_t rows, int cols, float* ws, int accumulate, hipStream_t s);
int launch_colsum_fold(const float* partial, float* out, int nrows, int cols, hipStream_t s);   // rows are per-block partial sums
// Deferred folds (round 5).  Every producer of a gradient leaves partial rows (split-K slabs of a weight gradient, per-block column sums
// of a LayerNorm / dropout / attention backward) that a small fold kernel turned into the final fp32 gradient right behind it: nine
// launches of 8-15 us per encoder layer, each with a grid too small to fill the chip.  Nothing inside the layer reads those gradients,
// so the folds of one layer are collected in a FoldBatch and run as ONE launch over a job table (the kernel argument itself) in front of
// the layer's bucket event.  A job keeps the summation order of the kernel it replaces (TALL = colsum_final_kernel<true>: sequential
// fp64 over the partial rows; WIDE = colsum_final_wide_kernel: 8 row lanes x 4 loads in flight), so results are bit-identical to the
// separate launches (tests/test_train_gpu.py::test_deferred_folds_do_not_change_results).  The producers' partial buffers must stay
// untouched until flush(): each deferred site has its own scratch (TrainState::site_slabs / site_ws).
struct FoldJob {
    const float* partial;
    float* out[3];        // WIDE: one target per column group.  TALL with unpack > 0: the three (unpack, unpack) kernels of a packed (unpack, 3 unpack) matrix
    int64_t ld;           // floats between partial rows
    int nchunks, cols;    // partial rows; columns (per group)
    int kind;             // 0 = TALL, 1 = WIDE
    int groups;           // WIDE: column groups sharing the partial rows (group g reads partial + g cols)
    int unpack;           // TALL: 0, or H of the packed q|k|v kernel
    int block0;           // first block of the job in the launch
};
constexpr int FOLD_MAX_JOBS = 12;
struct FoldTable {
    FoldJob j[FOLD_MAX_JOBS];
    int n;
};
struct FoldBatch {
    FoldTable tab;
    int nblocks = 0;
    double bytes = 0.0;
    FoldBatch() { tab.n = 0; }
    // false: the job does not fit the table or the 16-byte path (the caller then runs the separate fold as before)
    bool add_tall(const float* partial, int nchunks, int64_t cols, float* out, int unpack = 0, float* out1 = nullptr, float* out2 = nullptr);
    bool add_wide(const float* partial, int nchunks, int cols, int64_t ld, int groups, float* out0, float* out1 = nullptr, float* out2 = nullptr);
    int flush(hipStream_t s);
};
int64_t ln_bwd_ws_floats(int64_t rows, int C);
int launch_ln_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma,
                  float* dbeta, int64_t rows, int C, float eps, float* ws, hipStream_t s);
// dropout backward of the tensor in front of a residual add, folded into the LayerNorm backward that produces its gradient
struct LnDropTail {
    float p;
    uint64_t seed;
    uint32_t stream;
};
int launch_ln_bwd_x(const float* x, const float* gamma, const float* dy, float* dx, uint16_t* dx16 /* optional bf16 shadow */,
                    float* dgamma, float* dbeta, int64_t rows, int C, float eps, float* ws, hipStream_t s,
                    float* dxsum = nullptr /* optional: column sums of dx (C floats) */,
                    const float* residual = nullptr /* optional: dx = LN-backward(dy) + residual */,
                    const struct LnDropTail* tail = nullptr /* optional: dx16 / dxsum take dropout-backward(dx) instead of dx */,
                    struct FoldBatch* defer = nullptr /* the fold of ws joins this batch (ws then stays live until its flush) */);
int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float b1, float b2,
                float eps, hipStream_t s);
int launch_spec_aug_fwd(const float* x, const uint8_t* mask, const float* embed, float* y, int64_t rows, int H, hipStream_t s);
int launch_spec_aug_bwd(const float* dy, const uint8_t* mask, float* dx, float* dmasked, int64_t rows, int H, hipStream_t s);
// dst (Kin, Nout) = A16[R rows]^T B16[R rows], bf16 operands, fp32 accumulate: the leftover rows of a weight gradient
int launch_dw_tail_bf16(const uint16_t* A16, int64_t lda, const uint16_t* B16, int64_t ldb, float* dst, int R, int Kin, int Nout, hipStream_t s);
int launch_axpby(const float* a, const float* b, float* y, int64_t n, float alpha, float beta, hipStream_t s);
int launch_axpby_x(const float* a, const float* b, float* y, uint16_t* y16 /* optional bf16 shadow */, int64_t n, float alpha, float beta,
                   hipStream_t s);
int launch_mask_rows(const float* x, const int32_t* frame_len, float* y, int B, int T, int H, hipStream_t s);

// attention, training variants (attention.hip)
struct AttnTrain {
    float p;           // attention-probability dropout (encoder.py:42-44)
    uint64_t seed;
    uint32_t stream;
    float* lse;        // (B, heads, T) log-sum-exp of the (masked) scores, written by forward, read by backward
    // optional (bf16 kernels, p > 0): the keep decisions of the forward, one bit per (query, key), so that the backward kernels
    // read them instead of hashing every element twice more.  attention_keep_bits_words(B, T, heads) 32-bit words; layout in
    // attention_bf16.hip.  Null: the backward recomputes the hash (same bits).
    uint32_t* keep_bits = nullptr;
};
int64_t attention_keep_bits_words(int B, int T, int heads);
// bf16 matrix-pipe attention (attention_bf16.hip); taken by launch_attention* while the thread's precision is 1
bool attention_bf16_supported(int head_size);
// (the kernels read bf16 shadows of qkv / dctx; a null shadow is made from the fp32 tensor in per-stream scratch)
int launch_attention_fwd_bf16(const float* qkv, const uint16_t* qkv16, const int32_t* frame_len, float* ctx, uint16_t* ctx16, int B,
                              int T, int H, int heads, const AttnTrain* tr, hipStream_t s);
// colpart (optional): (attention_colpart_rows(B, T), 3H) per-block column sums of dqkv; summed over its rows they are the q|k|v
// bias gradient.  dqkv may be null when dqkv16 is given.
int launch_attention_bwd_bf16(const float* qkv, const uint16_t* qkv16, const int32_t* frame_len, const float* dctx, const uint16_t* dctx16,
                              float* dvec, float* dqkv, uint16_t* dqkv16 /* optional bf16 shadow */, int B, int T, int H, int heads,
                              const AttnTrain& tr, hipStream_t s, float* colpart = nullptr, const float* ctx = nullptr /* fp32 O: D computed inside */,
                              const uint16_t* ctx16 = nullptr /* ... or its bf16 copy (the forward's shadow): same D */);
int attention_colpart_rows(int B, int T);
int launch_attention_train_x(Profiler* prof, const float* qkv, const uint16_t* qkv16, const int32_t* frame_len, float* ctx, uint16_t* ctx16,
                             int B, int T, int H, int heads, const AttnTrain& tr, hipStream_t s);
int launch_attention_train(Profiler* prof, const float* qkv, const int32_t* frame_len, float* ctx, int B, int T,
                           int H, int heads, const AttnTrain& tr, hipStream_t s);
// dqkv (B, T, 3H) = gradient of the packed q|k|v given dctx (B, T, H); ctx is the forward output
// (bf16 kernels: qkv16 / dctx16 are the optional shadows of qkv / dctx, qkv may be null when qkv16 is given)
int launch_attention_bwd(Profiler* prof, const float* qkv, const int32_t* frame_len, const float* ctx,
                         const float* dctx, float* dqkv, float* dvec_ws, int B, int T, int H, int heads,
                         const AttnTrain& tr, hipStream_t s, uint16_t* dqkv16 = nullptr /* bf16 shadow of dqkv (bf16 kernels only) */,
                         const uint16_t* qkv16 = nullptr, const uint16_t* dctx16 = nullptr, float* colpart = nullptr /* see launch_attention_bwd_bf16 */,
                         const uint16_t* ctx16 = nullptr /* bf16 kernels: ctx / dctx may then be null (O and dO are read as bf16 only) */);

// positional conv, training variants (posconv.hip)
int launch_pos_conv_ex(Profiler* prof, const float* x, const float* wg, const float* bias,
                       const int32_t* frame_len, float* y, float* pre_act, int B, int T, int H, int K,
                       int groups, int act, int pad_left, int add_residual, hipStream_t s);
int launch_pos_conv_flip_regroup(const float* wg, float* wg_t, int K, int cg, int groups, hipStream_t s);
int launch_pos_conv_dw(Profiler* prof, const float* xz, const float* dc, float* dwg, float* ws, int B, int T,
                       int H, int K, int groups, hipStream_t s);
int64_t pos_conv_dw_ws_floats(int B, int T, int H, int K, int groups);
int launch_weight_norm_bwd(const float* wv, const float* wgain, const float* dwg, float* dwv, float* dwgain,
                           int K, int cg, int H, int groups, hipStream_t s);

}  // namespace w2v2
