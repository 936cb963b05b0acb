// Model state shared by the forward (w2v2_api.hip) and training (w2v2_train.hip) translation units.
#pragma once

#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

struct Param {
    std::string name;
    std::vector<int64_t> shape;
    int64_t numel = 0;
    float* dev = nullptr;
    bool set = false;
};
struct Act {
    float* ptr;
    int64_t shape[3];
};

struct w2v2_model {
    w2v2_config cfg;
    std::vector<Param> params;
    std::unordered_map<std::string, int> index;
    // derived tensors (w2v2_finalize)
    float* pos_wg = nullptr;                 // (groups, K, cg, og)
    std::vector<float*> qkv_w, qkv_b;        // per layer (H, 3H), (3H)
    bool finalized = false;
    int precision = 0;                       // 0 fp32 MFMA, 1 bf16 operands / fp32 accumulate (w2v2_set_precision)
    bool opt_shadows = true, opt_keep_acts = false;      // w2v2_set_option
    // activation workspace
    int ws_B = 0;
    int64_t ws_L = 0;
    std::vector<void*> allocs;
    std::map<std::string, Act> acts;
    std::vector<std::string> acts_skipped;   // stage outputs the LAST forward wrote only as bf16 (precision mode 1): no fp32 copy to read back
    std::vector<float*> conv;                // conv stack outputs
    std::vector<int> conv_T;
    float *conv0_ws = nullptr, *ln512 = nullptr, *proj = nullptr, *posout = nullptr;
    std::vector<float*> hs;                  // hidden states: encoder_in, layer0..N-1
    float *qkv = nullptr, *ctx = nullptr, *t0 = nullptr, *t1 = nullptr, *t2 = nullptr, *t3 = nullptr,
          *ffn = nullptr, *enc_out = nullptr;
    int32_t* frame_len = nullptr;
    // bf16 shadows (precision mode 1, inference forward; w2v2_api.hip::ensure_shadows).  Weight shadows are the
    // GEMM kernels transposed to (N, K); activation shadows are written by the producing kernels.
    bool sh_ready = false, w16_valid = false;
    std::unordered_map<const float*, uint16_t*> w16;      // (N, K): forward GEMMs
    std::unordered_map<const float*, uint16_t*> w16p;     // plain (K, N) bf16 copy = the (N, K) shadow of W^T: dX GEMMs
    std::vector<void*> sh_allocs, w16_allocs;
    w2v2::ShadowJob* shadow_jobs = nullptr;               // device table of the single-launch weight-shadow refresh
    int shadow_njobs = 0;
    bool shadow_jobs_train = false;                       // the table includes the plain copies (built after training state existed)
    std::vector<uint16_t*> conv16, hs16;
    uint16_t *qkv16 = nullptr;                            // q|k|v as the bf16 attention kernels read it (the fp32 copy is then not written)
    uint16_t *ln512_16 = nullptr, *ctx16 = nullptr, *t0_16 = nullptr, *t2_16 = nullptr, *ffn16 = nullptr, *enc16 = nullptr;
    // bf16 positional conv (precision mode 1; posconv.hip): kernel shadow (groups, og, K cg), pack scratch (B, G, T+K-1, cg)
    uint16_t *pos_w16 = nullptr, *pos_pack16 = nullptr;
    bool pos16_valid = false;
    // precision mode 2 (gemm_split.hip): three (N, K) bf16 planes per GEMM weight, built on first use, rebuilt after finalize
    struct SplitPlanes { uint16_t* p = nullptr; int64_t elems = 0; uint64_t epoch = 0; };
    std::unordered_map<const float*, SplitPlanes> w48;     // keyed by the fp32 matrix (a variable, or a transposed copy)
    uint64_t w48_epoch = 1;                                 // bumped whenever the variables change: entries re-split lazily
    // precision modes 2 / 3 on the plane-fed GEMM (gemm_split_sw.hip; w2v2_api.hip::w2v2_ensure_planes): the planes of every GEMM
    // operand, written by its producer (three bf16 planes or two fp16 planes per element: PlaneFmt), and the weights as that kernel's
    // LDS images, built on first use and re-split lazily after the variables change (w48_epoch).  Stream order lets one buffer serve
    // every layer: attention input | ctx | FFN input | FFN hidden.
    struct PlaneBuf { uint16_t* p = nullptr; int64_t plane = 0; };
    bool opt_planes = true;                                  // W2V2_OPT_SPLIT_PLANES
    bool opt_wgrad_stream = false;                           // W2V2_OPT_WGRAD_STREAM
    bool opt_defer_folds = true;                             // W2V2_OPT_DEFER_FOLDS
    int pl_fmt = -1, pl_B = 0;
    int64_t pl_L = 0;
    std::vector<void*> pl_allocs;
    std::vector<PlaneBuf> conv48;                            // conv-stack outputs 0 .. NC-2
    PlaneBuf ln512_48, attn_in48, ctx48, ffn_in48, ffn48;
    int* range_flag = nullptr;                               // sticky fp16 saturation flag (device)
    struct SplitImages { uint16_t* img = nullptr; float* scale_ws = nullptr; int64_t elems = 0; uint64_t epoch = 0; };
    std::unordered_map<const float*, SplitImages> wimg[2];   // [PlaneFmt], keyed by the fp32 matrix
    w2v2::Profiler* prof = nullptr;
    struct TrainState* train = nullptr;      // owned by w2v2_train.hip (null until the first training call)
    struct Comm* comm = nullptr;             // owned by comm.hip: the native RCCL communicator (null until w2v2_comm_init)

    float* P(const std::string& n) const {
        auto it = index.find(n);
        return it == index.end() ? nullptr : params[it->second].dev;
    }
};


// implemented in w2v2_api.hip
bool w2v2_shadows_enabled(const w2v2_model* m);                            // W2V2_OPT_BF16_SHADOWS (default on)
bool w2v2_conv_out_bf16_only(const w2v2_model* m, int i, bool sh);         // conv-stack output i is written only as bf16 this forward
bool w2v2_conv_ln_bf16_only(const w2v2_model* m, int i, bool sh);          // LayerNorm-mode extractor: LN + GELU output i only as bf16
bool w2v2_keep_activations(const w2v2_model* m);                           // W2V2_OPT_KEEP_ACTIVATIONS: also write the fp32 copies nothing reads
int w2v2_ensure_shadows(w2v2_model* m, int B, int T, hipStream_t s);
bool w2v2_pos_conv_bf16_ok(const w2v2_model* m);                          // precision 1 and a supported group shape
int w2v2_ensure_pos16(w2v2_model* m, int B, int T, hipStream_t s);       // kernel shadow + pack scratch     // allocate activation shadows, (re)build weight shadows
int w2v2_ensure_workspace(w2v2_model* m, int B, int64_t L);
// precision mode 2: the LDS-image bf16 planes of the (K, N) fp32 matrix `W` (built / refreshed on demand; gemm_split.hip)
int w2v2_split_planes(w2v2_model* m, const float* W, int K, int N, hipStream_t s, const uint16_t** planes);
// whether GEMM (M, N, K) x nbatch with this A should take the split kernel in the model's current precision mode
bool w2v2_use_split_gemm(const w2v2_model* m, const float* A, int64_t lda, int64_t strideA, int64_t ldb, int M, int N, int K, int nbatch);
// precision modes 2 / 3 with W2V2_OPT_SPLIT_PLANES: the plane buffers of a (B, L) forward in the mode's format; the LDS images (and,
// f16x2, the accumulator scale) of the (K, N) fp32 matrix W
int w2v2_ensure_planes(w2v2_model* m, int B, int64_t L, int fmt);
int w2v2_split_images(w2v2_model* m, const float* W, int K, int N, int fmt, hipStream_t s, const uint16_t** img, const float** out_scale);
// implemented in w2v2_train.hip
void w2v2_train_destroy(w2v2_model* m);
void w2v2_train_invalidate(w2v2_model* m);
// the contiguous runs of TRAINABLE slots (offset, numel; 16-byte aligned slots merge) inside gradient bucket k, and the flat buffer
int w2v2_train_trainable_runs(w2v2_model* m, int k, std::vector<std::pair<int64_t, int64_t>>* runs, float** grads);
// implemented in comm.hip
void w2v2_comm_free(w2v2_model* m);
