/*
 * w2v2.h -- C ABI of the MI355X-native Wav2Vec2 forward / CTC path.
 *
 * The reference (thevasudevgupta/gsoc-wav2vec2) has NO plugin / operator / FFI
 * interface: its hot path sits behind a plain Python package surface
 * (src/wav2vec2/__init__.py:1-4) and every arithmetic op is a TensorFlow call.
 * This header is therefore the boundary a maintainer of that package would
 * bind *instead of* TensorFlow: each entry point names the reference call it
 * replaces.  The Python host side (gsoc-wav2vec2_amd/wav2vec2) binds it with
 * ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every function returns 0 on success, a negative W2V2_E* code on failure;
 *     w2v2_last_error() returns a thread-local message.  No exceptions cross
 *     the ABI.
 *   - `*_dev` pointers are HIP device pointers (fp32 unless stated), owned by
 *     the caller; `*_host` pointers are host memory.  The model owns its
 *     weights and its activation workspace.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *     work is enqueued on it; nothing synchronises unless stated.
 *   - one model per stream at a time; the library is not thread-safe per model.
 *   - activations are channels-last (batch, time, channels), as in the
 *     reference; kernels are (K, C_in, C_out) / (in, out) -- the reference's
 *     TF checkpoint layout (src/convert_torch_to_tf.py:110-117).
 */
#ifndef W2V2_H_
#define W2V2_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define W2V2_MAX_CONV_LAYERS 16

enum {
    W2V2_OK = 0,
    W2V2_EINVAL = -1,     /* bad argument / shape / config           */
    W2V2_ENOTFOUND = -2,  /* unknown parameter or activation name    */
    W2V2_EHIP = -3,       /* a HIP runtime call failed               */
    W2V2_ESTATE = -4      /* call order (e.g. forward before all params are set) */
};

/* Mirrors Wav2Vec2Config (reference src/wav2vec2/config.py:6-60); only the
 * fields the forward / CTC arithmetic reads.  dropout, survival_prob and the
 * spec-augment fields are train-time host concerns. */
typedef struct w2v2_config {
    int32_t vocab_size;
    int32_t hidden_size;
    int32_t num_heads;
    int32_t num_layers;
    int32_t intermediate_size;
    int32_t num_conv_pos_embeddings;
    int32_t num_conv_pos_embedding_groups;
    int32_t num_conv_layers;
    int32_t filter_sizes[W2V2_MAX_CONV_LAYERS];
    int32_t kernal_sizes[W2V2_MAX_CONV_LAYERS];   /* sic: the reference's spelling is API */
    int32_t strides[W2V2_MAX_CONV_LAYERS];
    int32_t conv_bias;                    /* 0 / 1 */
    int32_t feature_extractor_norm_type;  /* 0 = "group", 1 = "layer"        */
    int32_t attention_norm_type;          /* 0 = "postnorm", 1 = "prenorm"   */
    int32_t is_gelu_approx;               /* 0 = exact erf GELU              */
    int32_t with_lm_head;                 /* 1 = Wav2Vec2ForCTC, 0 = Wav2Vec2Model */
    int32_t pad_id;                       /* CTC blank index                 */
    float   layer_norm_eps;
} w2v2_config;

typedef struct w2v2_model w2v2_model;

const char* w2v2_last_error(void);
const char* w2v2_version(void);
/* Free the library-owned per-(device, stream) scratch buffers (split-K slabs, CTC alpha / beta) of the calling thread's
 * current device; they are otherwise kept, grow-only, for the life of the process.  The device must be idle. */
int w2v2_release_scratch(void);

/* ---- model lifetime ------------------------------------------------------
 * Replaces Wav2Vec2Model.__init__ / Wav2Vec2ForCTC.__init__
 * (reference modeling.py:105-167, 217-233): allocates every variable on the
 * current HIP device (zero-filled). */
int  w2v2_create(const w2v2_config* cfg, w2v2_model** out);
void w2v2_destroy(w2v2_model* m);

/* Variable inventory, named as the reference's TF variables without the model
 * prefix and ":0" (convert_torch_to_tf.py:12-44), e.g.
 * "encoder/layers/3/attention/q_proj/kernel", "lm_head/bias". */
int w2v2_num_params(const w2v2_model* m);
int w2v2_param_info(const w2v2_model* m, int index, const char** name,
                    int64_t shape[4], int* rank);

/* Replaces model.load_weights / keras batch_set_value
 * (modeling.py:82, convert_torch_to_tf.py:121): copy one variable host->device.
 * `shape`/`rank` must match the inventory. */
int w2v2_set_param(w2v2_model* m, const char* name, const float* host_src,
                   const int64_t* shape, int rank);
/* Replaces model.variables[i].numpy() (convert_torch_to_tf.py:81). */
int w2v2_get_param(w2v2_model* m, const char* name, float* host_dst, int64_t numel);

/* Derives the tensors the kernels consume from the variables: the
 * weight-normalised positional kernel  l2_normalize(weight_v,[1,2])*weight_g
 * (tensorflow_addons.py:16-21) regrouped per conv group, and the packed
 * q|k|v projection.  Must be called after the last w2v2_set_param and before
 * w2v2_forward; enqueued on `stream`. */
int w2v2_finalize(w2v2_model* m, void* stream);

/* Frames out of the conv stack for `num_samples` input samples:
 * 1 + (len - k) // s per layer (modeling.py:202-204, losses.py:47-56). */
int64_t w2v2_num_frames(const w2v2_model* m, int64_t num_samples);

/* Arithmetic of the dense contractions (Conv1D layers 1..6 and every Dense, forward and backward):
 *   W2V2_PRECISION_FP32  v_mfma_f32_32x32x2_f32 on fp32 operands -- the reference's arithmetic (default);
 *   W2V2_PRECISION_BF16  operands rounded to bf16 (nearest-even) on the way into LDS, fp32 accumulation,
 *                        fp32 bias / GELU / residual / storage: what BASELINE configs "bf16 CTC fine-tune"
 *                        ask for (mixed precision; variables, optimizer state and activations stay fp32).
 *                        Attention (head size 64) takes bf16 q, k, v and probabilities on the same pipe with
 *                        fp32 scores / softmax / accumulation; the grouped positional conv runs as one batched
 *                        GEMM with bf16-rounded input and kernel.  In the TRAINING step of this mode four tensors that sit
 *                        between two roundings are also stored as bf16 (as a mixed_bfloat16 Dense would hand them on): the
 *                        FFN pre-activation (GELU and GELU' see bf16(u)), the gradient of the FFN hidden activation, the
 *                        attention output and its gradient (D = rowsum(dO o O) is taken from the bf16 values).  The
 *                        residual stream, LayerNorm inputs / outputs and every gradient of them stay fp32.
 *   W2V2_PRECISION_BF16X3  fp32 results from the bf16 matrix cores (forward, and the data-gradient GEMMs of the
 *                        training step): every fp32 operand is written
 *                        exactly as a sum of three bf16 terms and each fp32 product is evaluated as the six bf16 x bf16
 *                        products of order <= 2, accumulated in fp32 (the dropped terms are < 2^-23 of the product, below
 *                        the rounding of an fp32 running sum).  Same fp32 inputs, outputs and error level as
 *                        W2V2_PRECISION_FP32 -- logits within the same 1e-3 of the reference -- at the bf16 pipe's rate.
 *                        Attention (head size 64) takes the same route; shapes the split kernels do not take, the
 *                        positional conv, the weight-gradient GEMMs and the training attention stay on the fp32 MFMA.
 *                        In the inference forward the producers of the GEMM operands (conv0, LayerNorm, the GEMM and attention
 *                        epilogues) write the three planes themselves and the GEMMs stream them (gemm_split_sw.hip).
 *   W2V2_PRECISION_F16X2   fp32-grade results from the fp16 matrix cores at HALF the MFMA work of BF16X3 (inference forward):
 *                        an operand is the sum of TWO fp16 terms of x 2^e (22 significant bits; e = 4 for activations, chosen
 *                        from max |w| for a weight), a product keeps a0 b0 + a0 b1 + a1 b0.  The error per product (~2^-22) is
 *                        below what an fp32 running sum over K >= 64 products commits: measured GEMM error at or below the
 *                        fp32 MFMA kernel's, logits within the same bar.  Domain: |activation| < 4094 (a larger value
 *                        saturates and sets a sticky flag, w2v2_range_overflow).  GEMM shapes its kernel does not take, the
 *                        attention core and the training step run as in BF16X3.
 * Everything else (conv0 + GroupNorm, LayerNorm, softmax, CTC) is fp32 in all modes. */
#define W2V2_PRECISION_FP32 0
#define W2V2_PRECISION_BF16 1
#define W2V2_PRECISION_BF16X3 2
#define W2V2_PRECISION_F16X2 3
int w2v2_set_precision(w2v2_model* m, int32_t mode);
int w2v2_get_precision(const w2v2_model* m);

/* Per-model switches of the bf16 precision mode (no environment variable is read anywhere in the library):
 *   W2V2_OPT_BF16_SHADOWS (default 1)      0 = no bf16 operand shadows: every GEMM rounds its fp32 operands itself.  Same
 *                                          results bit for bit, slower; the parity tests flip it to prove exactly that.
 *   W2V2_OPT_KEEP_ACTIVATIONS (default 0)  1 = also write the fp32 copies of stage outputs whose only reader streams the bf16
 *                                          shadow, so w2v2_copy_activation can tap them (otherwise such a tap is an error).
 *   W2V2_OPT_SPLIT_PLANES (default 1)      precision modes BF16X3 / F16X2: 0 = no operand planes written by the producers; the
 *                                          forward GEMMs then load fp32 rows and split them in registers (gemm_split.hip, six
 *                                          bf16 products in both modes) -- the round-4 path, kept for A/B measurements.
 *   W2V2_OPT_WGRAD_STREAM (default 0)      bf16 training backward: 1 = the encoder layers' weight-gradient GEMMs run on a second,
 *                                          lower-priority HIP stream owned by the model, ordered against the caller's stream by
 *                                          events only (w2v2_train_backward returns with the caller's stream waiting for all of it;
 *                                          bucket events then come from that stream).  Same results bit for bit; measured neutral
 *                                          on one GPU (profiles/history.md 7.1), hence off.
 *   W2V2_OPT_DEFER_FOLDS (default 1)       training backward: the small reductions that finish an encoder layer's gradients (split-K
 *                                          slab sums of the four weight gradients, the q|k|v unpack, the LayerNorm / dropout /
 *                                          attention column-sum folds: nine launches per layer) run as ONE launch in front of the
 *                                          layer's bucket event.  Same summation order, same bits; 0 = one launch behind each
 *                                          producer (the round-4 behaviour, kept for the bit-identity test and A/B timing).
 * w2v2_get_option returns the value, or W2V2_EINVAL for an unknown option. */
#define W2V2_OPT_BF16_SHADOWS 0
#define W2V2_OPT_KEEP_ACTIVATIONS 1
#define W2V2_OPT_SPLIT_PLANES 2
#define W2V2_OPT_WGRAD_STREAM 3
#define W2V2_OPT_DEFER_FOLDS 4
int w2v2_set_option(w2v2_model* m, int32_t option, int32_t value);
int w2v2_get_option(const w2v2_model* m, int32_t option);
/* W2V2_PRECISION_F16X2: *flag = 1 if a forward since the last call met an activation outside fp16's range after scaling
 * (|x| >= 4094: its planes were saturated, the logits of that forward are NOT fp32-grade -- rerun in BF16X3 or FP32); clears the
 * flag.  Synchronises `stream`.  Always 0 in the other modes. */
int w2v2_range_overflow(w2v2_model* m, int32_t* flag, void* stream);

/* ---- the hot path --------------------------------------------------------
 * Replaces Wav2Vec2ForCTC.call / Wav2Vec2Model.call at training=False
 * (modeling.py:169-209, 239-255).
 *   wave_dev   (B, L) fp32 waveform, already normalised + padded by the caller
 *   mask_dev   (B, L) int32 0/1 attention mask, or NULL (base checkpoints)
 *   out_dev    (B, T, vocab) logits if with_lm_head else (B, T, hidden);
 *              T = w2v2_num_frames(L)
 * The first call for a new (B, L) allocates the activation workspace
 * (hipMalloc; not capturable); later calls with B*L no larger reuse it. */
int w2v2_forward(w2v2_model* m, const float* wave_dev, int32_t B, int64_t L,
                 const int32_t* mask_dev, float* out_dev, void* stream);

/* Replaces CTCLoss.call (losses.py:14-45) = tf.nn.ctc_loss(
 * logits_time_major=False, blank_index=pad_id) with the reference's length
 * convention supplied by the caller:
 *   logits_dev       (B, T, V) fp32
 *   labels_dev       (B, U) int32, label_length_dev (B) int32  (count of labels != pad)
 *   logit_length_dev (B) int32 (the reference passes the full T for every row)
 *   nll_dev          (B) fp32 per-sample negative log-likelihood (not divided,
 *                    not reduced: division_factor and the SUM are the caller's)
 *   grad_logits_dev  (B, T, V) fp32 d(sum_b nll_b)/d logits, or NULL to skip */
int w2v2_ctc_loss(const float* logits_dev, int32_t B, int32_t T, int32_t V,
                  const int32_t* labels_dev, int32_t U,
                  const int32_t* label_length_dev, const int32_t* logit_length_dev,
                  int32_t blank, float* nll_dev, float* grad_logits_dev, void* stream);
/* The same with CTCLoss.call's conventions evaluated on the device, so that the caller issues no tensor arithmetic around the call
 * (the training step: no framework kernels between the forward and the backward; capturable):
 *   label_length = count of labels != blank per row (losses.py:32-33); logit_length = logit_length_all for every row (losses.py:29-30:
 *   the full frame count); grad_logits_dev = d(sum_b nll_b) / d logits / division_factor (losses.py:45: a DIVISION, as the reference
 *   computes it -- multiplying by a rounded reciprocal differs by an ulp for factors that are not powers of two);
 *   loss_sum_dev (optional device scalar) = sum_b (nll_b / division_factor), added in row order (Keras Reduction.SUM, losses.py:6). */
int w2v2_ctc_loss_fused(const float* logits_dev, int32_t B, int32_t T, int32_t V, const int32_t* labels_dev, int32_t U,
                        int32_t logit_length_all, int32_t blank, float division_factor, float* nll_dev, float* grad_logits_dev,
                        float* loss_sum_dev, void* stream);

/* ---- the training step (reference src/main.py:136-259) --------
 * Replaces what Keras' train_step does around the forward: training-mode forward, backward of every
 * trainable variable, Adam.  Postnorm (base) and prenorm (robust / xlsr) transformers; the conv feature
 * extractor has no backward (the reference freezes it, main.py:234-237) -- mark it non-trainable first.
 *
 * w2v2_train_forward: Wav2Vec2ForCTC.call(training=True): Dropout(p) at every Dropout layer
 *   (feature_extractor.py:95; encoder.py:42-44,118,128,270; modeling.py:253) with masks from a
 *   counter-based hash of (seed, site, element) -- regenerated in the backward (the bf16 attention keeps its
 *   decisions as one bit per score for its own backward); the probability is realised to 2^-16;
 *   spec_mask_host (B*T bytes, or NULL): frames replaced by masked_spec_embed (spec_augment.py:119-127;
 *   the span sampling itself is host-side numpy in the reference and stays on the host here);
 *   sd_keep_host (num_layers floats of 0/1, or NULL): StochasticDepth's one Bernoulli draw per layer
 *   call (tensorflow_addons.py:381).  Saves what the backward needs.
 * w2v2_train_backward: given d loss / d logits (B, T, V) fills the flat gradient buffer (zero for
 *   frozen variables).  w2v2_grad_buffer exposes that buffer for the data-parallel all-reduce (SUM).
 * w2v2_adam_step: Keras Adam, p -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps), then re-derives the
 *   tensors w2v2_finalize builds. */
int w2v2_set_trainable(w2v2_model* m, const char* name_prefix, int trainable);
/* The whole flag vector in one call: flags[i] != 0 marks variable i of the inventory (w2v2_param_info order) trainable;
 * n must equal w2v2_num_params.  This is what `model.trainable = ...` / `layer.trainable = ...` push after walking the
 * Keras-style layer tree (reference src/main.py:210,234-237 flips whole layers). */
int w2v2_set_trainable_flags(w2v2_model* m, const uint8_t* flags, int32_t n);
int w2v2_train_forward(w2v2_model* m, const float* wave_dev, int32_t B, int64_t L, const int32_t* mask_dev,
                       const uint8_t* spec_mask_host, const float* sd_keep_host, float dropout_p,
                       uint64_t seed, float* logits_dev, void* stream);
int w2v2_train_backward(w2v2_model* m, const float* grad_logits_dev, void* stream);
int w2v2_grad_buffer(w2v2_model* m, float** dev_ptr, int64_t* numel);
/* Gradient buckets for overlapping the data-parallel all-reduce with the backward (the reference leaves this to
 * tf.distribute; main.py:156,192).  Bucket k is a contiguous slice [offset, offset + numel) of the flat gradient buffer;
 * buckets are numbered in the order w2v2_train_backward completes them: 0 = lm_head, 1 .. N = encoder layers N-1 .. 0,
 * N+1 = everything in front of layer 0 (positional conv, feature projection, ...).  They tile the buffer exactly.
 * w2v2_train_bucket_wait makes `stream` wait until bucket k of the LAST enqueued backward is final -- and for nothing
 * else, so a collective enqueued behind it runs under the rest of the backward. */
int w2v2_train_num_buckets(const w2v2_model* m);
int w2v2_train_bucket(w2v2_model* m, int32_t k, int64_t* offset, int64_t* numel);
int w2v2_train_bucket_wait(w2v2_model* m, int32_t k, void* stream);
/* Slot [offset, offset + numel) of one variable in the flat gradient / Adam buffers (inventory order, 16-byte aligned slots):
 * lets the caller send only the trainable runs of a bucket (the reference's payload excludes the frozen conv stack). */
int w2v2_grad_slot(w2v2_model* m, const char* name, int64_t* offset, int64_t* numel);

/* ---- native data-parallel collective (RCCL over xGMI; gsoc-wav2vec2_amd/csrc/comm.hip) ------------------------------------------
 * The gradient SUM of the reference's MirroredStrategy step (src/main.py:148-156,192: one replica per device; :198-200: the loss is
 * divided by the GLOBAL batch, so the cross-replica SUM of the gradients is the global-mean gradient) issued by the library itself:
 * one process per GPU, one communicator per model.  RCCL is bound at run time (dlopen librccl.so.1; a copy the process already holds
 * is reused), so hosts without it only lose these entry points (W2V2_ESTATE with the reason in w2v2_last_error).
 *   w2v2_comm_unique_id    rank 0 draws the rendezvous id (W2V2_COMM_ID_BYTES bytes) and hands it to the other ranks by any channel
 *                          the host has (a file, MPI, a TCP store: bytes, not a torch type);
 *   w2v2_comm_init         every rank: ncclCommInitRank on the CURRENT device + the library's own high-priority communication stream;
 *   w2v2_allreduce_bucket  in-place SUM of gradient bucket k's TRAINABLE runs (frozen slots are zero on every rank and stay home:
 *                          90,195,104 + 768 elements = 360.8 MB for wav2vec2-base in stage 2) on the communication stream,
 *                          which first waits for that bucket's completion event of the last enqueued backward -- and for nothing
 *                          else: call it for k = 0 .. w2v2_train_num_buckets - 1 right after w2v2_train_backward and the upper layers'
 *                          collectives run under the lower layers' backward.  algo 0 = ncclAllReduce; 1 = ncclReduceScatter +
 *                          ncclAllGather over the run's world-divisible body (+ an all-reduce of the < world-element tail);
 *   w2v2_allreduce_finish  `stream` (the optimizer's) waits for everything enqueued on the communication stream; *payload_bytes
 *                          (may be NULL) = bytes reduced since the previous finish;
 *   w2v2_allreduce_num_runs / _run   the (offset, numel) runs a bucket sends (tests; equal to wav2vec2/dist.py::trainable_ranges).
 * Not thread-safe; every rank must make the same calls in the same order (RCCL's rule). */
#define W2V2_COMM_ID_BYTES 128
int w2v2_comm_unique_id(uint8_t* id_out, int32_t nbytes);
int w2v2_comm_init(w2v2_model* m, const uint8_t* unique_id, int32_t nbytes, int32_t rank, int32_t world);
int w2v2_comm_info(const w2v2_model* m, int32_t* rank, int32_t* world, int32_t* rccl_version);
int w2v2_comm_destroy(w2v2_model* m);
int w2v2_allreduce_num_runs(w2v2_model* m, int32_t k, int32_t* count);
int w2v2_allreduce_run(w2v2_model* m, int32_t k, int32_t i, int64_t* offset, int64_t* numel);
int w2v2_allreduce_bucket(w2v2_model* m, int32_t k, int32_t algo);
int w2v2_allreduce_finish(w2v2_model* m, void* stream, int64_t* payload_bytes);
/* Adam's moment buffers (device, flat, same layout as the gradient buffer): what a training checkpoint must carry besides
 * the variables and the step count (the reference's ModelCheckpoint writes a TF checkpoint, training_utils.py:38-45). */
int w2v2_adam_buffers(w2v2_model* m, float** m_dev, float** v_dev, int64_t* numel);
/* A fresh optimizer: zero both moment buffers.  The reference creates a new tf.keras.optimizers.Adam for each of its two
 * stages (src/main.py:213,240); the moments live in the model's training state here, so whoever starts a new optimizer
 * (a new Trainer, or a checkpoint without moments) calls this. */
int w2v2_adam_reset(w2v2_model* m, void* stream);
int w2v2_get_grad(w2v2_model* m, const char* name, float* host_dst, int64_t numel, void* stream);
/* Which per-layer activations the LAST training forward kept only as bf16 (precision mode BF16 on the shadow paths; their fp32 buffers
 * are then not allocated at all), and the bytes of shape-dependent training workspace currently allocated.  Introspection for the tests:
 * a step that silently fell back to the fp32 activations (twice the element-wise traffic) must not pass unnoticed.
 *   *mask: bit 0 = q|k|v, bit 1 = attention output, bit 2 = FFN hidden activation, bit 3 = FFN pre-activation u (bf16 in half a buffer),
 *          bit 4 = (prenorm) the in-layer LayerNorm outputs */
#define W2V2_TRAIN_BF16_QKV 1
#define W2V2_TRAIN_BF16_CTX 2
#define W2V2_TRAIN_BF16_FFN 4
#define W2V2_TRAIN_BF16_U 8
#define W2V2_TRAIN_BF16_LN 16   /* prenorm only: the outputs of the two LayerNorms inside a layer (they feed GEMMs only) */
int w2v2_train_storage(const w2v2_model* m, int32_t* mask, int64_t* workspace_bytes);
int w2v2_adam_step(w2v2_model* m, float lr, float beta1, float beta2, float eps, int64_t step, void* stream);

/* ---- introspection (parity tests, profiling) -----------------------------
 * Stage activations of the LAST forward, by name: "conv0".."conv6",
 * "projection", "encoder_in", "layer0".."layerN-1", "encoder_out".  Copies
 * device -> host after synchronising `stream`. */
int w2v2_activation_info(const w2v2_model* m, const char* name, int64_t shape[3]);
int w2v2_copy_activation(w2v2_model* m, const char* name, float* host_dst,
                         int64_t numel, void* stream);

/* Per-kernel-family timing with HIP events on the launch stream.
 * enable=1 starts recording (every launch is bracketed by an event pair);
 * w2v2_profile_read synchronises and returns, for family `index`, its name,
 * launch count, total milliseconds, algorithmic FLOPs and algorithmic bytes
 * summed over the recorded launches; w2v2_profile_reset clears the record. */
int w2v2_profile_enable(w2v2_model* m, int enable);
/* Restrict the event pairs to some families (bit i = family index i of w2v2_profile_read; 0 = all):
 * an event pair costs ~7 us of stream time, so timing ONLY the dominant family keeps the timed region
 * of a benchmark within 1 % of the un-instrumented run. */
int w2v2_profile_families(w2v2_model* m, uint32_t family_mask);
/* Sampling: only every stride-th launch of a family gets the event pair (default 1 = every launch).  With a stride coprime to
 * the number of launches per layer the samples rotate through all shapes; averages stay unbiased, the tax drops by the stride.
 * w2v2_profile_seen: how many launches of the family were issued since the last reset, sampled or not. */
int w2v2_profile_sampling(w2v2_model* m, int32_t stride);
int w2v2_profile_seen(w2v2_model* m, int index, int64_t* launches);
/* KERNEL launches enqueued for the family since the last w2v2_profile_reset, by any model of the process: an op-level call
 * counted by w2v2_profile_seen may enqueue several kernels (main + tail-tile kernels of one GEMM, partial + final reductions);
 * this is the count a rocprofv3 --kernel-trace of the same steps shows.  Counted whether or not profiling is enabled. */
int w2v2_profile_kernel_launches(w2v2_model* m, int index, int64_t* launches);
int w2v2_profile_num_families(void);
int w2v2_profile_read(w2v2_model* m, int index, const char** name, int64_t* launches,
                      double* total_ms, double* flops, double* bytes);
int w2v2_profile_reset(w2v2_model* m);

/* Shader-clock probe (measurement aid; no reference counterpart).  Enqueues on `stream` a one-wave kernel that samples the
 * shader-cycle counter and the constant-rate wall clock, spins `spin_us` microseconds of wall time (0 < spin_us <= 200000) and
 * samples both again into dev_out4 = {cycles0, wall0, cycles1, wall1} (device memory, 4 x uint64).  *wall_clock_khz receives the
 * wall counter's rate.  Shader MHz while it ran = (cycles1 - cycles0) / (wall1 - wall0) * wall_clock_khz / 1000.  Launched on
 * a side stream while a workload runs on another, it reports the clock the chip holds UNDER that workload (MI355X clocks to its
 * power budget), which bench.py prints beside the nominal-peak roofline fraction. */
int w2v2_clock_probe(void* stream, int32_t spin_us, uint64_t* dev_out4, int32_t* wall_clock_khz);

/* ---- individual operators (each is one hot-path kernel family) -----------
 * Exposed so every kernel is parity-tested on its own against the oracle. */

/* C[z] = act(A[z] @ Bm + bias) + residual[z]      (tf.keras.layers.Dense /
 * Conv1D-as-implicit-GEMM; feature_extractor.py:31-37, encoder.py:15-18,99-104)
 *   A  (M, K) row-major with leading dimension lda (elements) -- an overlapping
 *      lda < K window view is how strided Conv1D is expressed;
 *   Bm (K, N) row-major, ldb;  C (M, N), ldc;  bias (N) or NULL;
 *   residual (M, N) with ldc or NULL (added AFTER the activation);
 *   act: 0 none, 1 exact GELU, 2 tanh GELU;
 *   batch z in [0, nbatch): A += z*strideA, C/residual += z*strideC. */
int w2v2_op_gemm(const float* A_dev, int64_t lda, int64_t strideA,
                 const float* B_dev, int64_t ldb,
                 float* C_dev, int64_t ldc, int64_t strideC,
                 const float* bias_dev, const float* residual_dev,
                 int32_t M, int32_t N, int32_t K, int32_t nbatch, int32_t act, void* stream);

/* Precision of the w2v2_op_* calls issued by the calling thread from now on (per-kernel tests of the bf16
 * mode); model-level calls take the model's own setting (w2v2_set_precision) and restore this one. */
int w2v2_op_set_precision(int32_t mode);

/* Same contract, operands rounded to bf16 / fp32 accumulate (W2V2_PRECISION_BF16's GEMM). */
int w2v2_op_gemm_bf16(const float* A_dev, int64_t lda, int64_t strideA,
                      const float* B_dev, int64_t ldb,
                      float* C_dev, int64_t ldc, int64_t strideC,
                      const float* bias_dev, const float* residual_dev,
                      int32_t M, int32_t N, int32_t K, int32_t nbatch, int32_t act, void* stream);

/* The same contraction fed from bf16 SHADOWS, the form the model's forward and data-gradient GEMMs run in precision mode bf16:
 *   A16 (M, K) bf16 rows lda elements apart (overlapping rows = strided Conv1D, batch stride strideA), B16 = the (N, K) bf16
 *   shadow of the (K, N) kernel; C fp32 and / or C16 bf16 (either may be NULL).  Both operands stream HBM / L2 -> LDS by DMA.
 * variant: 0 = the kernel the library would pick for the shape, 1 = the 128 x 128-tile kernel (gemm_bf16.hip), 2 = the 128 x 256
 * software-pipelined kernel with two 4-wave blocks per CU (gemm_bf16_sw.hip; needs N % 256 == 0, K % 64 == 0, K >= 192).  All
 * variants produce identical bits (tests/test_ops_gpu.py). */
int w2v2_op_gemm_bf16_shadows(const uint16_t* A16_dev, int64_t lda, int64_t strideA, const uint16_t* B16_nk_dev, float* C_dev,
                              uint16_t* C16_dev, int64_t ldc, int64_t strideC, const float* bias_dev, const float* residual_dev,
                              int32_t M, int32_t N, int32_t K, int32_t nbatch, int32_t act, int32_t variant, void* stream);

/* C_z = act(A_z B + bias) + residual_z with fp32 operands split exactly into three bf16 terms each and six bf16 MFMA
 * products per fp32 product (W2V2_PRECISION_BF16X3's GEMM; B dense (K, N), split into planes inside the call).
 * N % 256 == 0, K % 32 == 0, lda % 4 == 0, 16-byte aligned A. */
int w2v2_op_gemm_split(const float* A_dev, int64_t lda, int64_t strideA, const float* B_dev,
                       float* C_dev, int64_t ldc, int64_t strideC,
                       const float* bias_dev, const float* residual_dev,
                       int32_t M, int32_t N, int32_t K, int32_t nbatch, int32_t act, void* stream);

/* The same contraction with BOTH operands pre-split into 16-bit planes, the form the model's forward GEMMs run in precision modes
 * bf16x3 and f16x2 (gemm_split_sw.hip: software-pipelined 128 x 256 tiles, operands HBM / L2 -> LDS by DMA).  fmt:
 *   W2V2_PLANES_BF16X3  three bf16 planes, x = p0 + p1 + p2 exactly; six MFMA products per fp32 product;
 *   W2V2_PLANES_F16X2   two fp16 planes of x * 2^e (activations: e = 4; a weight: e chosen from max |w|); three products.
 *   w2v2_op_split_planes   x (n fp32) -> planes, plane p at planes + p * plane_stride (what the producing kernels of the model write
 *                          next to / instead of their fp32 output);
 *   w2v2_op_split_weight   Bm (K, N) fp32 -> the kernel's weight images, plane_count * K * N 16-bit elements (done once per variable
 *                          update).  f16x2: scale_ws_dev = two device words, [0] work space, [1] receives the fp32 scale
 *                          1 / (2^e_w * 2^4) that the GEMM must be given as out_scale_dev; bf16x3: NULL;
 *   w2v2_op_gemm_split_planes   A as planes (rows lda elements apart, overlapping rows = strided Conv1D, batch stride strideA, planes
 *                          planeA elements apart), B as images; the result as fp32 C (+ residual) or -- C NULL -- as the planes of
 *                          act(A B + bias) at C16 + p * planeC for the next GEMM.  range_flag_dev (int, may be NULL): set to 1 when an
 *                          f16x2 plane output saturates fp16 (|x| >= 4094).  N % 256 == 0, K % 64 == 0, lda % 8 == 0, 16-byte aligned planes. */
#define W2V2_PLANES_BF16X3 0
#define W2V2_PLANES_F16X2 1
int w2v2_op_split_planes(const float* x_dev, uint16_t* planes_dev, int64_t plane_stride, int64_t n, int32_t fmt, int32_t* range_flag_dev,
                         void* stream);
int w2v2_op_split_weight(const float* B_dev, uint16_t* images_dev, float* scale_ws_dev, int32_t K, int32_t N, int32_t fmt, void* stream);
int w2v2_op_gemm_split_planes(int32_t fmt, const uint16_t* A16_dev, int64_t planeA, int64_t lda, int64_t strideA,
                              const uint16_t* B_images_dev, const float* out_scale_dev,
                              float* C_dev, uint16_t* C16_dev, int64_t planeC, int64_t ldc, int64_t strideC,
                              const float* bias_dev, const float* residual_dev,
                              int32_t M, int32_t N, int32_t K, int32_t nbatch, int32_t act, int32_t* range_flag_dev, void* stream);

/* Self-check of the branch-free GELU forms the bf16x3 GEMM epilogues use (csrc/common.h: erf_select, tanh_select): evaluates them
 * and the device library's erff / tanhf on all 2^32 float bit patterns; mismatches_dev[0] / [1] receive the number of patterns whose
 * results differ in any bit (NaNs compare equal).  Both must be 0: the forms are drop-in for the fp32 path's GELU. */
int w2v2_op_check_select_forms(uint64_t* mismatches_dev, void* stream);

/* The weight-gradient form of the same GEMM: C_z (M, N) = A_z^T B_z with A given TRANSPOSED, (K, M) fp32 row-major with
 * row stride lda (>= M), and per-batch strides on A, B and C (split-K: batch z covers rows [z K, (z+1) K) of both
 * operands and writes slab z).  bf16-rounded operands, fp32 accumulation.  K % 64 == 0, M % 4 == 0, N % 4 == 0. */
int w2v2_op_gemm_bf16_at(const float* At_dev, int64_t lda, int64_t strideA,
                         const float* B_dev, int64_t ldb, int64_t strideB,
                         float* C_dev, int64_t ldc, int64_t strideC,
                         int32_t M, int32_t N, int32_t K, int32_t nbatch, void* stream);

/* The same weight gradient from bf16 copies of both operands in their natural layouts: slab z (Kin, Nout) = X_z^T dY_z over rows
 * [z rows_per_slab, (z + 1) rows_per_slab) of x16 (rows, Kin) and dy16 (rows, Nout), uint16 bf16 bit patterns, row-major.  `rows`
 * need not fill the last slab (nor be a multiple of the 64-row K tile): rows past it count as zero, every slab must own at least
 * one row.  Kin % 128 == 0, Nout % 128 == 0, rows_per_slab % 64 == 0, 16-byte aligned operands.  (The fine-tune step's dW GEMMs
 * in W2V2_PRECISION_BF16: tf.GradientTape's kernel gradient of Dense, src/main.py:198.)
 * variant: 0 = the kernel the library would pick, 1 = the 128 x 128-tile transposing-read kernel, 2 = the 128 x 256 software-pipelined
 * kernel in its transposed form (whole 64-row K tiles, Nout % 256 == 0, >= 192 rows per slab); identical bits.  Variant 2 also takes
 * UNEVEN slabs, the form the training step uses to cut B T rows into any number of slabs: rows = rows_per_slab nslabs + 64 e with
 * 0 < e < nslabs gives each of the first e slabs 64 rows more, slab z then starting at row z rows_per_slab + 64 min(z, e).
 * Variant 3 = variant 2 plus the caller's promise that dy16 has one more row, index `rows`, holding zeros: `rows` may then end
 * inside the last K tile (its missing rows are read from that zero row and from x16's last row) -- what the training step does
 * with its own dY shadows at B T = 23984 (480000-sample utterances). */
int w2v2_op_weight_grad_bf16(const uint16_t* x16_dev, const uint16_t* dy16_dev, float* slabs_dev,
                             int64_t rows, int32_t Kin, int32_t Nout, int32_t rows_per_slab, int32_t nslabs, int32_t variant,
                             void* stream);

/* CRC-32C (Castagnoli; TFRecord and TensorFlow-checkpoint checksums, tensorflow/core/lib/hash/crc32c.h: crc32c::Extend) of `n` host
 * bytes continuing from `crc` (0 to start).  Host-only helper for the package's checkpoint / record I/O: no device work. */
uint32_t w2v2_crc32c_extend(uint32_t crc, const void* data_host, uint64_t n);

/* y = LN(x) * gamma + beta over the last axis, optional GELU after
 * (tf.keras.layers.LayerNormalization(axis=-1); act as above). rows x C. */
int w2v2_op_layer_norm(const float* x_dev, float* y_dev, const float* gamma_dev,
                       const float* beta_dev, int64_t rows, int32_t C, float eps,
                       int32_t act, void* stream);

/* Layer 0 of the feature extractor in "group" mode: Conv1D(C_in=1, K, stride,
 * valid) -> GroupNormalization(groups=C) = per-(sample, channel) statistics
 * over time -> GELU, without materialising the un-normalised conv output
 * (feature_extractor.py:31-47,54-59; tensorflow_addons.py:207-231).
 *   wave (B, L); kernel (K, 1, C); bias (C) or NULL; out (B, T0, C);
 *   stats_ws: caller scratch of w2v2_conv0_ws_floats(B, L, K, stride, C) floats.
 * norm_mode: 0 = group-norm + activation, 1 = conv(+bias) only, 2 = conv -> LayerNormalization over the C channels of each frame
 * -> activation in one pass (the "layer" configs, feature_extractor.py:40-50): a frame's channel mean and variance follow from
 * its 10 input samples and a 10 x 10 moment matrix of the kernel, so the un-normalised conv output is never written.  Mode 2
 * needs gamma, beta and stats_ws; geometries other than K = 10, stride 5, C % 4 == 0 run mode 1 + w2v2_op_layer_norm inside.
 * act as in w2v2_op_layer_norm (modes 0 and 2). */
int64_t w2v2_conv0_ws_floats(int32_t B, int64_t L, int32_t K, int32_t stride, int32_t C);
int w2v2_op_conv0(const float* wave_dev, const float* kernel_dev, const float* bias_dev,
                  const float* gamma_dev, const float* beta_dev, float* out_dev,
                  float* stats_ws_dev, int32_t B, int64_t L, int32_t K, int32_t stride,
                  int32_t C, float eps, int32_t norm_mode, int32_t act, void* stream);

/* Effective positional kernel: l2_normalize(weight_v, axes [1,2]) * weight_g,
 * regrouped to (groups, K, C_in/groups, C_out/groups)
 * (tensorflow_addons.py:16-21; encoder.py:168-175). */
int w2v2_op_weight_norm_regroup(const float* weight_v_dev, const float* weight_g_dev,
                                float* wg_dev, int32_t K, int32_t cg, int32_t H,
                                int32_t groups, void* stream);

/* y = xz + GELU(grouped_conv_same(xz) + bias), xz = x with frames >=
 * frame_len[b] zeroed (encoder.py:253,265; PositionalConvEmbedding
 * encoder.py:177-181: pad K/2 both sides, drop the last frame for even K).
 *   x, y (B, T, H); wg from w2v2_op_weight_norm_regroup; frame_len (B) or NULL. */
int w2v2_op_pos_conv(const float* x_dev, const float* wg_dev, const float* bias_dev,
                     const int32_t* frame_len_dev, float* y_dev, int32_t B, int32_t T,
                     int32_t H, int32_t K, int32_t groups, int32_t act, void* stream);

/* Multi-head self-attention context (TransformerAttention.get_context,
 * encoder.py:34-47, with the q pre-scale of :28): per (batch, head)
 * softmax((q*scale) k^T + mask) v, mask = -10000 on keys >= frame_len[b].
 *   qkv (B, T, 3H) packed q|k|v rows;  ctx (B, T, H). Scores are never
 *   materialised. */
int w2v2_op_attention(const float* qkv_dev, const int32_t* frame_len_dev, float* ctx_dev,
                      int32_t B, int32_t T, int32_t H, int32_t num_heads, void* stream);

/* frame_len[b] = conv-stack length arithmetic applied to sum(mask[b, :])
 * (modeling.py:201-204). */
int w2v2_op_frame_lengths(const int32_t* mask_dev, int32_t* frame_len_dev, int32_t B,
                          int64_t L, const int32_t* kernal_sizes, const int32_t* strides,
                          int32_t num_layers, void* stream);

/* ---- training operators (parity-tested on their own) ----------------------------------------- */

/* LayerNormalization backward: dx, dgamma, dbeta from x (pre-norm input), gamma and dy.
 * ws: scratch of w2v2_ln_bwd_ws_floats(rows, C) floats. */
int64_t w2v2_ln_bwd_ws_floats(int64_t rows, int32_t C);
int w2v2_op_layer_norm_bwd(const float* x_dev, const float* gamma_dev, const float* dy_dev, float* dx_dev,
                           float* dgamma_dev, float* dbeta_dev, int64_t rows, int32_t C, float eps,
                           float* ws_dev, void* stream);

/* Attention with dropout on the probabilities (encoder.py:42-44) + the saved row statistic (B, heads, T),
 * and its backward: dqkv (B, T, 3H) from dctx (B, T, H).  dvec_ws: (B, heads, T) scratch.
 * lse_dev is what the forward leaves for the backward of the SAME precision mode: the natural log-sum-exp of the (masked, scaled)
 * scores under W2V2_PRECISION_FP32; minus the log2-sum-exp2 (= -lse * log2 e, the kernels' own exponent units: P = exp2(S c + lse2)
 * is then one fused multiply-add and one v_exp_f32 per score in the backward) under W2V2_PRECISION_BF16. */
int w2v2_op_attention_train(const float* qkv_dev, const int32_t* frame_len_dev, float* ctx_dev, float* lse_dev,
                            int32_t B, int32_t T, int32_t H, int32_t num_heads, float dropout_p,
                            uint64_t seed, uint32_t stream_id, void* stream);
int w2v2_op_attention_bwd(const float* qkv_dev, const int32_t* frame_len_dev, const float* ctx_dev,
                          const float* lse_dev, const float* dctx_dev, float* dqkv_dev, float* dvec_ws_dev,
                          int32_t B, int32_t T, int32_t H, int32_t num_heads, float dropout_p,
                          uint64_t seed, uint32_t stream_id, void* stream);

/* y = dropout(act(x)) [+ residual]: keep iff (hash16(seed, stream_id, index) ^ 0x8000) >= floor(p 2^16) -- the 16-bit value read as
 * a signed number --, kept values / (1 - p)  (csrc/train.h, wav2vec2/variables.py::dropout_keep: the same integer function). */
int w2v2_op_dropout(const float* x_dev, const float* residual_dev, float* y_dev, int64_t n, int32_t act,
                    float p, uint64_t seed, uint32_t stream_id, void* stream);

/* t1 = dropout(x) + residual  and  y = LayerNorm(t1) over the last axis (C % 4 == 0), one pass over the rows: the training forward's
 * "x + drop(attn(x))" followed by the layer's next LayerNorm (encoder.py:116-124).  Bit-identical to w2v2_op_dropout (act 0) followed by
 * w2v2_op_layer_norm (act 0) on the same operands; y16_dev (optional) receives the nearest-even bf16 copy of y. */
int w2v2_op_layer_norm_dropout(const float* x_dev, const float* residual_dev, float* t1_dev, float* y_dev, uint16_t* y16_dev,
                               const float* gamma_dev, const float* beta_dev, int64_t rows, int32_t C, float eps, float p, uint64_t seed,
                               uint32_t stream_id, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* W2V2_H_ */
