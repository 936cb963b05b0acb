// Shared host/device helpers for the gfx950 Wav2Vec2 path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include <string>

#include "../../include/w2v2.h"

namespace w2v2 {

// ---- error plumbing (no exceptions across the C ABI) ----------------------
void set_error(const char* fmt, ...);

#define W2V2_HIP_CHECK(expr)                                                            \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            ::w2v2::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                              __FILE__, __LINE__);                                      \
            return W2V2_EHIP;                                                           \
        }                                                                               \
    } while (0)

#define W2V2_REQUIRE(cond, ...)                \
    do {                                       \
        if (!(cond)) {                         \
            ::w2v2::set_error(__VA_ARGS__);    \
            return W2V2_EINVAL;                \
        }                                      \
    } while (0)

// ---- tuning knobs -----------------------------------------------------------
// The shipping library has NO environment reads: every knob is its measured default, folded at compile time.  The tools-only
// build (`python build.py --tuning` -> lib/libw2v2_tuning.so, -DW2V2_TUNING) reads the same names from the environment for
// the sweeps and timing ablations recorded under profiles/; tests and bench.py never load it.
#ifdef W2V2_TUNING
int tune_int(const char* name, int dflt);      // getenv + atoi, read on every call (w2v2_api.hip)
#else
constexpr int tune_int(const char*, int dflt) { return dflt; }
#endif

// ---- tile order of the large-tile GEMM kernels inside an XCD's run of tiles (round 6) ----
// Every GEMM kernel hands each XCD (blockIdx % 8) one contiguous run of the linear tile order, so that tiles in flight together on
// an XCD share operand panels in ITS L2 (4 MB).  With the linear order row-major (N fastest), the ~64 tiles an XCD has in flight are
// 64 / tiles_n rows x ALL tiles_n columns: on the wide shapes (N = 3072 / 4096: 12 - 32 column tiles) that is 2 - 5 row panels of A
// beside the WHOLE of B, streamed through L2 again for every such set -- PMC (profiles/hbm_traffic_configs.json, round 5): the bf16
// q|k|v / FFN-up kernels of the large model fetched 577 MB per launch for 57 MB of operands, the fp32 256 x 128 instances 960 MB.
// Grouped order: the run walks groups of `gm` tile rows, column-major inside a group; gm is the number of A row panels
// (BM x K elements each) that fit a 3.25-MB share of the L2, so a group's A panels stay resident while B streams past them ONCE per
// group instead of once per 64 tiles; the tiles in flight form a gm x (64 / gm) patch.  gm = 0 keeps the linear order (narrow N, the
// weight-gradient form with its own shorter-dimension-fastest rule).  Bijective for any tile count (the last group is shorter).
__device__ __forceinline__ void grouped_tile(int t, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
    const int per = gm * tiles_n, grp = t / per, first = grp * gm;
    const int rows = min(gm, tiles_m - first), tl = t - grp * per;
    tn = tl / rows;
    tm = first + (tl - tn * rows);
}
// host side: group height for a launch (0 = linear order).  a_panel_bytes = BM x K x element size; inflight = tiles an XCD runs at once
inline int tile_group_rows(int tiles_m, int tiles_n, int64_t a_panel_bytes, int inflight) {
    const int forced = tune_int("W2V2_TILE_GROUP", -1);
    if (forced >= 0) return forced > tiles_m ? tiles_m : forced;
    if (tiles_m < tiles_n || tiles_n < 6) return 0;      // narrow outputs: inflight / tiles_n rows x all columns is already the patch
    int64_t gm = (int64_t)3407872 / (a_panel_bytes > 0 ? a_panel_bytes : 1);      // 3.25 MB of the 4-MB L2 for the resident A panels
    const int need = (inflight + tiles_n - 1) / tiles_n;      // never fewer rows than the linear order has in flight
    if (gm < need) gm = need;
    if (gm < 2) gm = 2;
    // an XCD's run covers tiles_m / 8 rows: split them into equal groups no taller than that (B streams past once per group)
    const int rows_xcd = (tiles_m + 7) / 8;
    if (gm < rows_xcd) {
        const int ngroups = (int)((rows_xcd + gm - 1) / gm);
        gm = (rows_xcd + ngroups - 1) / ngroups;
    }
    return (int)(gm > tiles_m ? tiles_m : gm);
}

// ---- kernel families (one row each in profiles/ and in the roofline) ------
enum Family {
    FAM_CONV0_STATS = 0,   // conv0 recompute + per-(sample,channel) sum/sumsq   (HBM: wave read)
    FAM_CONV0_APPLY,       // conv0 recompute + GroupNorm + GELU + single write  (HBM: 100.76 MB/utt write)
    FAM_GEMM,              // fp32 MFMA GEMM / implicit-GEMM conv / Dense        (MFMA f32)
    FAM_GEMM_BF16,         // same contractions, bf16 operands / fp32 accumulate (MFMA bf16; precision mode 1)
    FAM_GEMM_SPLIT,        // same contractions, fp32 operands as 3 bf16 terms each, 6 MFMA products (precision mode 2)
    FAM_LAYERNORM,         // row LayerNorm (+GELU)                              (HBM)
    FAM_POSCONV,           // grouped positional conv, MFMA 16x16x4 f32          (MFMA f32)
    FAM_ATTENTION,         // fused QK^T-softmax-PV, MFMA 32x32x2 f32            (MFMA f32)
    FAM_CTC,               // CTC alpha/beta                                     (latency)
    FAM_MISC,              // frame lengths, weight-norm regroup, packing, spec-augment, residual adds, transposes
    // training-step-only families (train_kernels.hip, shadow.hip): HBM-bound element-wise passes and reductions
    FAM_DROPOUT,           // dropout forward / backward (+ GELU / GELU', + fused column sums)   (HBM / VALU)
    FAM_REDUCE,            // column sums: bias gradients, weight-gradient slab folds            (HBM)
    FAM_LN_BWD,            // LayerNorm backward (+ its partial-sum fold)                        (HBM)
    FAM_OPTIMIZER,         // Adam (one launch over a chunk table) + the weight-shadow refresh   (HBM)
    FAM_COUNT
};
const char* family_name(int f);

// plane formats of the pre-split GEMM operands (gemm_split_sw.hip): three bf16 planes, x = p0 + p1 + p2 exactly (precision mode
// bf16x3), or two fp16 planes of x S (precision mode f16x2; helpers further down)
enum PlaneFmt { PF_BF16X3 = 0, PF_F16X2 = 1 };
constexpr int plane_count(int fmt) { return fmt == PF_F16X2 ? 2 : 3; }
// where a producing kernel leaves the planes of its output for the consumer GEMM (same element layout as the fp32 output; null p = none)
struct PlaneOut {
    uint16_t* p = nullptr;
    int64_t plane = 0;         // elements between planes
    int fmt = PF_BF16X3;
    int* range_flag = nullptr; // f16x2: sticky overflow flag in device memory
};

// Kernel launches actually enqueued, per family (an op-level call may enqueue several kernels: main + tail tiles, a partial
// and a final reduction ...).  Every launch in csrc/ goes through W2V2_LAUNCH; the family is the innermost live ProfScope's
// (FAM_MISC outside any).  Process-wide counters, cleared by profiler_reset; read by w2v2_profile_kernel_launches.
extern thread_local int tl_launch_family;
void note_kernel_launch();
int64_t kernel_launches(int family);
#define W2V2_LAUNCH(...)                  \
    do {                                  \
        ::w2v2::note_kernel_launch();     \
        hipLaunchKernelGGL(__VA_ARGS__);  \
    } while (0)

// A launch record sink.  When `enabled`, every launch wrapper brackets its
// kernel with an event pair on the launch stream and logs algorithmic
// flops/bytes; reading synchronises the events.
struct Profiler;
Profiler* profiler_create();
void profiler_destroy(Profiler*);
void profiler_enable(Profiler*, bool on);
void profiler_set_mask(Profiler*, unsigned family_mask);   // bit f = record family f; 0 = all
void profiler_set_stride(Profiler*, int stride);           // every stride-th launch of a family gets the event pair
int64_t profiler_seen(const Profiler*, int family);        // launches since the last reset, sampled or not
int64_t profiler_kernel_launches(const Profiler*, int family);      // kernels enqueued (process-wide counter) since THIS profiler's last reset
bool profiler_enabled(const Profiler*);
void profiler_reset(Profiler*);
// returns a token (>=0) to pass to profiler_end, or -1 when disabled
int profiler_begin(Profiler*, int family, double flops, double bytes, hipStream_t s);
void profiler_end(Profiler*, int token, hipStream_t s);
int profiler_read(Profiler*, int family, int64_t* launches, double* ms, double* flops, double* bytes);

// (a scope opened inside another one -- a launcher with its own scope called from a launcher that has one -- neither brackets its
//  kernels again nor re-labels their launches: time and launch counts stay with the OUTER family, nothing is counted twice)
extern thread_local int tl_prof_depth;
struct ProfScope {
    Profiler* p;
    int tok;
    hipStream_t s;
    int outer_family;
    ProfScope(Profiler* p_, int family, double flops, double bytes, hipStream_t s_)
        : p(p_), tok((p_ && tl_prof_depth == 0) ? profiler_begin(p_, family, flops, bytes, s_) : -1), s(s_), outer_family(tl_launch_family) {
        if (tl_prof_depth++ == 0) tl_launch_family = family;
    }
    ~ProfScope() {
        if (--tl_prof_depth == 0) tl_launch_family = outer_family;
        if (tok >= 0) profiler_end(p, tok, s);
    }
};

// The profiler of the model whose training step is running on this thread: the training-only kernels (train_kernels.hip,
// shadow.hip) take no Profiler argument; their launchers read it from here (null outside w2v2_train_* / w2v2_adam_step).
extern thread_local Profiler* tl_step_prof;
struct StepProfScope {
    Profiler* outer;
    explicit StepProfScope(Profiler* p) : outer(tl_step_prof) { tl_step_prof = p; }
    ~StepProfScope() { tl_step_prof = outer; }
};

// ---- operator launchers (defined one per .hip file) ------------------------
// All return 0 / negative W2V2_E*; `prof` may be null.
int launch_gemm(Profiler* prof, const float* A, int64_t lda, int64_t strideA, const float* B,
                int64_t ldb, float* C, int64_t ldc, int64_t strideC, const float* bias,
                const float* residual, int M, int N, int K, int nbatch, int act, hipStream_t s);

// precision mode 1: operands rounded to bf16 on the way into LDS, fp32 accumulate (gemm_bf16.hip)
int launch_gemm_bf16(Profiler* prof, const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb,
                     int64_t strideB, float* C, int64_t ldc, int64_t strideC, const float* bias,
                     const float* residual, int M, int N, int K, int nbatch, int act, hipStream_t s);
// fp32 GEMM as six bf16 MFMA products per fp32 product (gemm_split.hip, precision mode 2): A fp32 (M, K) rows lda apart,
// the weight pre-split by launch_split_weight into three bf16 planes stored as the kernel's LDS images (3 N K elements).
bool gemm_split_supported(const float* A, int64_t lda, int64_t strideA, int M, int N, int K);
int launch_split_weight(const float* w, uint16_t* planes, int K, int N, hipStream_t s);
int launch_gemm_split(Profiler* prof, const float* A, int64_t lda, int64_t strideA, const uint16_t* planes, float* C, int64_t ldc,
                      int64_t strideC, const float* bias, const float* residual, int M, int N, int K, int nbatch, int act,
                      hipStream_t s);

// ... and with BOTH operands pre-split (gemm_split_sw.hip), in plane format `fmt` (PlaneFmt below: three bf16 planes = precision mode
// bf16x3, two fp16 planes = precision mode f16x2): the activation as row-major planes `planeA` elements apart (written by its
// producer; launch_split_planes is the unfused form), the weight as that kernel's LDS images (launch_split_weight_sw: plane_count x N x K
// elements; the f16x2 format also needs two scratch words, whose second is the `out_scale` the GEMM reads); the result as fp32 (C,
// + residual) or as the planes of (acc + bias -> act) for the next GEMM (C16, planeC).  range_flag: sticky overflow flag of f16x2.
bool gemm_split_sw_ok(const uint16_t* A16, int64_t planeA, int64_t lda, int64_t strideA, int M, int N, int K);
int launch_split_weight_sw(const float* w, uint16_t* img, int K, int N, int fmt, void* scratch, hipStream_t s);
int launch_split_planes(const float* x, uint16_t* planes, int64_t plane, int64_t n, int fmt, int* range_flag, hipStream_t s);
int launch_gemm_split_sw(Profiler* prof, int fmt, const uint16_t* A16, int64_t planeA, int64_t lda, int64_t strideA, const uint16_t* Bimg,
                         const float* out_scale, float* C, uint16_t* C16, int64_t planeC, int64_t ldc, int64_t strideC, const float* bias,
                         const float* residual, int M, int N, int K, int nbatch, int act, int* range_flag, hipStream_t s);

// selftest.hip: counts the float patterns on which erf_select / tanh_select differ from the device library's erff / tanhf
int launch_check_select_forms(unsigned long long* mismatches_dev /* [2] */, hipStream_t s);

// Optional bf16 shadows of the operands (precision mode 1).  A shadow holds nearest-even bf16 roundings of the fp32
// tensor -- exactly what the kernel would round to itself -- so using one changes speed, never results.
//   A16: same shape / strides (in elements) as A;   B16: B TRANSPOSED, [N][K] with row stride ldb16 (0 = K);
//   C16: bf16 copy of the output for the consumer GEMM (C itself may then be null: the fp32 store is skipped).
struct GemmShadows {
    const uint16_t* A16 = nullptr;
    const uint16_t* B16 = nullptr;
    uint16_t* C16 = nullptr;
    int64_t ldb16 = 0;
    // A is stored TRANSPOSED: element (m, k) at A[k * lda + m] (fp32).  The weight-gradient GEMM dW = X^T dY passes the
    // activation X itself this way, so no transposed copy of X is made in precision mode 1.
    bool transA = false;
    // two-level batch z = zo * zmod + zi for a grouped conv run as one GEMM launch: A(16) advances with z (strideA), B16 and
    // bias with zi (strideB16, strideBias), C / residual with zo * strideC2 + zi * strideC.  zmod = 0: plain batch.
    int zmod = 0;
    int64_t strideB16 = 0, strideC2 = 0, strideBias = 0;
    int64_t strideB2 = 0;        // fp32 B with a two-level batch: B advances with zo * strideB2 + zi * strideB
    bool overlapA = false;       // transposed A whose rows overlap (lda < M): the packed positional-conv input
    // SRC 7 (fp32 transposed A) only: the K dimension is a concatenation of segments of `kseg` rows (a multiple of 64) that lie segA (A) /
    // segB (B) elements apart -- the positional-conv kernel gradient contracts over the frames of SEVERAL samples in one accumulator
    // (segment = sample) instead of writing one slab per sample.  kseg = 0: contiguous K.
    int kseg = 0;
    int64_t segA = 0, segB = 0;
    // transposed-A form only: also write the column sums of B over each batch's K rows to colsum[z * strideCS + n]
    // (dW = X^T dY has the bias gradient 1^T dY for free: dY is in registers while it is staged)
    float* colsum = nullptr;
    int64_t strideCS = 0;
    // transposed-A form with BOTH operands from bf16 shadows in their natural row-major layouts: A16 = bf16 copy of the
    // (K, M) activation (same lda / strideA as A), B16p = bf16 copy of the (K, N) gradient (same ldb / strideB as B).  Both then
    // stream into LDS by DMA as they lie in memory ([k][m] and [k][n] rows) and the k-contiguous MFMA fragments come out of the
    // transposing LDS read ds_read_b64_tr_b16 (gemm_bf16.hip: gemm_bf16_tr_kernel).  No column sums in this form (they are sums
    // of the UNROUNDED gradient): `colsum` must be null.
    const uint16_t* B16p = nullptr;
    // (that form only) the K dimension really has validK rows over all batches, K * nbatch >= validK > K * (nbatch - 1): rows past
    // it are read as zero.  0 = K * nbatch.  Lets dW = X^T dY run over B T = 23984 rows (T = 1499) without a leftover-row pass.
    int64_t validK = 0;
    // transposed-A shadow form on the 128 x 256 kernel: the first `kextra` batches (slabs) own one K tile (64 rows) more than K says
    // and every slab starts where the previous one ends -- K dimension = 64 (nbatch K / 64 + kextra) rows, strideA / strideB = K rows.
    // Lets the rows be cut into ANY number of slabs (7 x 72 tiles = 504 of the 512 block slots instead of 6 x 72 = 432).
    int kextra = 0;
    // ... and when validK ends inside the last K tile of the last slab, that kernel takes the ragged form too IF the caller keeps the
    // row of B16p just past the end (row validK) all-zero: the missing rows are then read from there (and from A's last row).
    bool b_zero_row = false;
    int force_kernel = 0;        // forward form with both shadows: 0 = by shape, 1 = the 128 x 128 kernel, 2 = the 128 x 256 software-pipelined one
};
// 128 x 256 software-pipelined form, two 4-wave blocks per CU (gemm_bf16_sw.hip): same arithmetic, identical bits
bool gemm_bf16_sw_ok(int M, int N, int K, int64_t lda, int64_t ldb16, int64_t strideA);
int launch_gemm_bf16_sw(const uint16_t* A16, int64_t lda, int64_t strideA, const uint16_t* B16, int64_t ldb16, float* C, uint16_t* C16,
                        int64_t ldc, int64_t strideC, const float* bias, const float* residual, int M, int N, int K, int nbatch, int act,
                        hipStream_t s);
bool gemm_bf16_swtr_ok(int M, int N, int K, int64_t lda, int64_t ldb, int64_t strideA, int64_t strideB);
int launch_gemm_bf16_swtr(const uint16_t* A16, int64_t lda, int64_t strideA, const uint16_t* B16, int64_t ldb, int64_t strideB, float* C,
                          int64_t ldc, int64_t strideC, int M, int N, int K, int nbatch, hipStream_t s, int kextra = 0, int krag = 0);
int launch_gemm_bf16_x(Profiler* prof, const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb,
                       int64_t strideB, float* C, int64_t ldc, int64_t strideC, const float* bias,
                       const float* residual, int M, int N, int K, int nbatch, int act, const GemmShadows& x,
                       hipStream_t s);
// launch_gemm / launch_gemm_ex route to launch_gemm_bf16 while the calling thread's precision is 1.  The
// API entry points set it from the model for the duration of one call (PrecisionScope).
void gemm_set_precision(int mode);
int gemm_get_precision();
struct PrecisionScope {
    int prev;
    explicit PrecisionScope(int mode) : prev(gemm_get_precision()) { gemm_set_precision(mode); }
    ~PrecisionScope() { gemm_set_precision(prev); }
};
int launch_gemm_ex(Profiler* prof, const float* A, int64_t lda, int64_t strideA, const float* B,
                   int64_t ldb, int64_t strideB, float* C, int64_t ldc, int64_t strideC, const float* bias,
                   const float* residual, int M, int N, int K, int nbatch, int act, hipStream_t s);

int launch_layer_norm(Profiler* prof, const float* x, float* y, const float* gamma,
                      const float* beta, int64_t rows, int C, float eps, int act, hipStream_t s);

int launch_layer_norm_x(Profiler* prof, const float* x, float* y, const float* gamma, const float* beta, int64_t rows,
                        int C, float eps, int act, uint16_t* y16 /* optional bf16 shadow of y */, hipStream_t s,
                        const PlaneOut* planes = nullptr /* optional planes of y (precision modes bf16x3 / f16x2; C % 4 == 0) */);

int64_t conv0_ws_floats(int B, int64_t L, int K, int stride, int C);
int launch_conv0(Profiler* prof, const float* wave, const float* kernel, const float* bias,
                 const float* gamma, const float* beta, float* out, float* ws, int B, int64_t L,
                 int K, int stride, int C, float eps, int norm_mode, int act, hipStream_t s);

int launch_conv0_x(Profiler* prof, const float* wave, const float* kernel, const float* bias,
                   const float* gamma, const float* beta, float* out, uint16_t* out16 /* optional bf16 shadow */, float* ws,
                   int B, int64_t L, int K, int stride, int C, float eps, int norm_mode, int act, hipStream_t s,
                   const PlaneOut* planes = nullptr /* optional planes of the output (the K = 10, stride 5, C % 4 == 0 kernel only) */);
// fp32 -> bf16 (nearest even): plain copy, and [K][N] -> [N][K] transpose (GEMM weight shadows)
int launch_to_bf16(const float* x, uint16_t* y, int64_t n, hipStream_t s);
// one 64 x 64 tile of one weight for the single-launch shadow refresh (shadow.hip): w (K, N) fp32 -> wt (N, K) bf16 and / or
// plain (K, N) bf16 (either may be null)
struct ShadowJob {
    const float* w;
    uint16_t* wt;
    uint16_t* plain;
    int K, N, k0, n0;
};
int launch_weight_shadows_multi(const ShadowJob* jobs_dev, int njobs, hipStream_t s);
int launch_transpose_to_bf16(const float* w, uint16_t* wt, int K, int N, hipStream_t s);
int launch_transpose_to_bf16_batched(const float* w, uint16_t* wt, int K, int N, int nbatch, hipStream_t s);   // nbatch dense (K, N) matrices

int launch_weight_norm_regroup(Profiler* prof, const float* wv, const float* wg, float* out, int K,
                               int cg, int H, int groups, hipStream_t s);
int launch_pos_conv(Profiler* prof, const float* x, const float* wg, const float* bias,
                    const int32_t* frame_len, float* y, int B, int T, int H, int K, int groups,
                    int act, hipStream_t s);

// precision mode 1: the grouped positional conv as one batched bf16 GEMM (posconv.hip)
int64_t pos_conv_bf16_pack_elems(int B, int T, int H, int K);
int launch_pos_conv_weight_shadow(const float* wg, uint16_t* w16, int K, int cg, int groups, hipStream_t s);
int launch_pos_conv_bf16(Profiler* prof, const float* x, const uint16_t* w16, const float* bias, const int32_t* frame_len,
                         float* y, float* pre_act, uint16_t* pack16, float* xz_ws, int B, int T, int H, int K, int groups,
                         int act, int pad_left, int add_residual, hipStream_t s);

// kernel gradient of the positional conv as a batched transposed-A GEMM (precision mode 1, T % 64 == 0):
//   dwg (groups, K, cg, og) = sum over samples and frames; pack32: (B, G, 64 ceil(T/64)+K-1, cg) fp32 scratch, slabs: B * K*cg*H fp32 scratch
int launch_pos_conv_dw_bf16(Profiler* prof, const float* xz, const float* dc, float* dwg, float* pack32, float* slabs, float* red_ws,
                            int B, int T, int H, int K, int groups, hipStream_t s, float* dc_pad = nullptr /* (B, 64 ceil(T/64), H): T % 64 != 0 */);

int launch_attention(Profiler* prof, const float* qkv, const int32_t* frame_len, float* ctx, int B,
                     int T, int H, int heads, hipStream_t s);

// grow-only device scratch per (purpose, stream), owned by the library (shadow.hip)
enum ScratchSlot { SCRATCH_SPLITK = 0, SCRATCH_CTC = 1, SCRATCH_QKV16 = 2, SCRATCH_DCTX16 = 3 };
int stream_scratch(int slot, hipStream_t s, size_t bytes, void** out);
int stream_scratch_release();       // frees the calling device's scratch buffers
// one layer's q | k | v projections <-> the packed (H, 3H) kernel and (3H) bias (shadow.hip); unpack skips null targets
int launch_qkv_pack(float* packed_w, float* packed_b, const float* const w[3], const float* const b[3], int H, hipStream_t s);
int launch_qkv_pack_layers(float* const* packed_w, float* const* packed_b, const float* const* w, const float* const* b, int layers, int H, hipStream_t s);
int launch_qkv_unpack(const float* packed_w, const float* packed_b, float* const w[3], float* const b[3], int H, hipStream_t s);
bool attention_bf16_supported(int head_size);   // attention_bf16.hip: head size 64
bool attention_split_supported(int head_size);  // attention_split.hip (precision mode 2): head size 64
int launch_attention_split(const float* qkv, const int32_t* frame_len, float* ctx, int B, int T, int H, int heads, hipStream_t s,
                           const PlaneOut* planes = nullptr /* optional planes of ctx; ctx itself may then be null */,
                           int fmt = PF_BF16X3 /* PF_F16X2: two fp16 terms / three products per contraction (precision mode 3) */,
                           int* range_flag = nullptr /* f16x2: sticky flag for q / k / v beyond fp16's scaled range */);
// qkv16: optional bf16 shadow of qkv (precision mode 1 with head size 64 reads ONLY it; qkv may then be null.  Without it that
// kernel rounds qkv into scratch first).  ctx may be null when ctx16 is given.
int launch_attention_x(Profiler* prof, const float* qkv, const uint16_t* qkv16, const int32_t* frame_len, float* ctx, int B, int T, int H,
                       int heads, uint16_t* ctx16 /* optional bf16 shadow of ctx (bf16 kernel only) */, hipStream_t s,
                       const PlaneOut* planes = nullptr /* optional planes of ctx (split kernel only: precision modes 2 / 3, head size 64) */);

int launch_frame_lengths(Profiler* prof, const int32_t* mask, int32_t* frame_len, int B, int64_t L,
                         const int32_t* ks, const int32_t* ss, int nl, hipStream_t s);

int launch_ctc(Profiler* prof, const float* logits, int B, int T, int V, const int32_t* labels,
               int U, const int32_t* label_len, const int32_t* logit_len, int blank, float* nll,
               float* grad, hipStream_t s);
// ... with the reference's conventions evaluated on the device (losses.py:29-45): label_len null = count of labels != blank,
// logit_len null = uniform_len frames for every row, the gradient / grad_div (division_factor), loss_sum (optional device
// scalar) = sum_b nll[b] / grad_div.  One call, no host-side tensor arithmetic around it.
int launch_ctc_x(Profiler* prof, const float* logits, int B, int T, int V, const int32_t* labels, int U, const int32_t* label_len,
                 const int32_t* logit_len, int uniform_len, int blank, float grad_div, float* nll, float* grad, float* loss_sum, hipStream_t s);

// ---- device helpers ---------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ float gelu_erf(float x) {
    // tf.nn.gelu(approximate=False): 0.5 x (1 + erf(x / sqrt(2)))
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// The device library's erff (ROCm 7.2 ocml, __ocml_erf_f32) without its branch: the |x| < 1 polynomial and the 1 - exp(-q(|x|))
// form are BOTH evaluated -- the same operations in the same order as the library's two arms -- and the result selected, so every
// input gives the library's bits (tools/erf_exhaustive.hip compares all 2^32 patterns on the GPU; tests/test_ops_gpu.py runs it).
// In a wave of 64 GELU inputs both arms execute anyway; what the select form removes is the control flow, which around 128
// accumulator registers makes the allocator spill (gemm_split_sw.hip: 570 bytes of scratch per lane with the branch, 0 without).
__device__ __forceinline__ float erf_select(float x) {
    const float ax = fabsf(x), t = x * x;
    float p = __builtin_fmaf(t, -0x1.268bc2p-11f, 0x1.420828p-8f);
    p = __builtin_fmaf(t, p, -0x1.b5937p-6f);
    p = __builtin_fmaf(t, p, 0x1.ce077cp-4f);
    p = __builtin_fmaf(t, p, -0x1.81266p-2f);
    p = __builtin_fmaf(t, p, 0x1.06ebap-3f);
    const float small = __builtin_fmaf(ax, p, ax);
    float q = __builtin_fmaf(ax, 0x1.1d3156p-16f, -0x1.8d129p-12f);
    q = __builtin_fmaf(ax, q, 0x1.f9a6d2p-9f);
    q = __builtin_fmaf(ax, q, -0x1.8c3164p-6f);
    q = __builtin_fmaf(ax, q, 0x1.b4e9c8p-4f);
    q = __builtin_fmaf(ax, q, 0x1.4515fap-1f);
    q = __builtin_fmaf(ax, q, 0x1.078e5p-3f);
    q = __builtin_fmaf(ax, q, ax);
    const float large = 1.0f - expf(-q);
    return copysignf(ax < 1.0f ? small : large, x);
}
// ... and the library's tanhf the same way (|x| < 0.625: odd polynomial; else 1 - 2 / (exp(2 |x|) + 1))
__device__ __forceinline__ float tanh_select(float x) {
    const float ax = fabsf(x), t = x * x;
    float p = __builtin_fmaf(t, -0x1.758e7ap-8f, 0x1.521192p-6f);
    p = __builtin_fmaf(t, p, -0x1.b8389cp-5f);
    p = __builtin_fmaf(t, p, 0x1.110704p-3f);
    p = __builtin_fmaf(t, p, -0x1.555532p-2f);
    const float small = __builtin_fmaf(t, ax * p, ax);
    const float large = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(expf(ax * 2.0f) + 1.0f), 1.0f);
    return copysignf(ax < 0.625f ? small : large, x);
}
__device__ __forceinline__ float gelu_tanh_select(float x) {     // == gelu_tanh(x), bit for bit
    const float c = 0.79788456080286535588f;  // sqrt(2/pi)
    return 0.5f * x * (1.0f + tanh_select(c * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float gelu_erf_select(float x) {      // == gelu_erf(x), bit for bit
    return 0.5f * x * (1.0f + erf_select(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_tanh(float x) {
    const float c = 0.79788456080286535588f;  // sqrt(2/pi)
    return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
}
// exact GELU 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz-Stegun 7.1.26: |erf error| < 1.5e-7 absolute, ~14 VALU
// ops against ~35 for erff.  Used by the bf16 kernel only (FAST_GELU): its error is a smooth, i.e. BIASED, function of x,
// and through 19 GELU layers that bias moved the fp32 CTC loss of the 246000-sample fixture from 5e-3 to 1.7e-2 off the
// fp64 reference (logits 7.4e-5 -> 8.1e-5) -- invisible next to bf16 rounding, not acceptable for the fp32 path.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
    const float erf_abs = fmaf(-p * t, e, 1.0f);             // erf(|x| / sqrt 2)
    return 0.5f * x + 0.5f * fabsf(x) * erf_abs;             // 0.5 x (1 + sign(x) erf(|x| / sqrt 2))
}

// gelu_erf_fast on two values at once (v_pk_mul_f32 / v_pk_fma_f32 where the ISA has packed forms; rcp and exp2 stay scalar)
using f32x2_t = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2_t gelu_erf_fast2(f32x2_t x) {
    const f32x2_t ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2_t z = ax * f32x2_t{0.70710678118654752440f, 0.70710678118654752440f};
    const f32x2_t d = __builtin_elementwise_fma(f32x2_t{0.3275911f, 0.3275911f}, z, f32x2_t{1.0f, 1.0f});
    const f32x2_t t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2_t q = __builtin_elementwise_fma(f32x2_t{1.061405429f, 1.061405429f}, t, f32x2_t{-1.453152027f, -1.453152027f});
    q = __builtin_elementwise_fma(q, t, f32x2_t{1.421413741f, 1.421413741f});
    q = __builtin_elementwise_fma(q, t, f32x2_t{-0.284496736f, -0.284496736f});
    q = __builtin_elementwise_fma(q, t, f32x2_t{0.254829592f, 0.254829592f});
    const f32x2_t zz = z * z * f32x2_t{-1.44269504088896340736f, -1.44269504088896340736f};
    const f32x2_t e = {__builtin_amdgcn_exp2f(zz[0]), __builtin_amdgcn_exp2f(zz[1])};
    const f32x2_t erf_abs = __builtin_elementwise_fma(-(q * t), e, f32x2_t{1.0f, 1.0f});
    const f32x2_t half = {0.5f, 0.5f};
    return __builtin_elementwise_fma(half * ax, erf_abs, half * x);
}

// act: 0 none, 1 exact GELU (erff), 2 tanh GELU, 3 exact GELU through gelu_erf_fast (element-wise kernels in precision mode 1)
__device__ __forceinline__ float apply_act(float x, int act) {
    return act == 1 ? gelu_erf(x) : (act == 2 ? gelu_tanh(x) : (act == 3 ? gelu_erf_fast(x) : x));
}
// two fp32 -> one dword of two bf16, nearest even (gfx950 v_cvt_pk_bf16_f32; no builtin in ROCm 7.2)
__device__ __forceinline__ unsigned pack_bf16_rne(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// exact three-term bf16 split of fp32 values (precision mode bf16x3): x = p0 + p1 + p2 with p0 = bf16(x), p1 = bf16(x - p0),
// p2 = x - p0 - p1 -- both subtractions are exact in fp32 and the last remainder fits bf16's 8 significant bits.
// Four values -> one dword pair per plane (element 0 in the low half of dword 0).
using f32x4_t = __attribute__((ext_vector_type(4))) float;
using u32x2_t = __attribute__((ext_vector_type(2))) unsigned;
__device__ __forceinline__ void split3_pack4(const f32x4_t& x, u32x2_t& p0, u32x2_t& p1, u32x2_t& p2) {
    p0[0] = pack_bf16_rne(x[0], x[1]);
    p0[1] = pack_bf16_rne(x[2], x[3]);
    f32x4_t r;
    r[0] = x[0] - __uint_as_float(p0[0] << 16);
    r[1] = x[1] - __uint_as_float(p0[0] & 0xffff0000u);
    r[2] = x[2] - __uint_as_float(p0[1] << 16);
    r[3] = x[3] - __uint_as_float(p0[1] & 0xffff0000u);
    p1[0] = pack_bf16_rne(r[0], r[1]);
    p1[1] = pack_bf16_rne(r[2], r[3]);
    r[0] -= __uint_as_float(p1[0] << 16);
    r[1] -= __uint_as_float(p1[0] & 0xffff0000u);
    r[2] -= __uint_as_float(p1[1] << 16);
    r[3] -= __uint_as_float(p1[1] & 0xffff0000u);
    p2[0] = pack_bf16_rne(r[0], r[1]);
    p2[1] = pack_bf16_rne(r[2], r[3]);
}
// one value -> its three bf16 bit patterns
__device__ __forceinline__ void split3_one(float x, uint16_t& p0, uint16_t& p1, uint16_t& p2) {
    const unsigned h0 = pack_bf16_rne(x, 0.f) & 0xffffu;
    const float r1 = x - __uint_as_float(h0 << 16);
    const unsigned h1 = pack_bf16_rne(r1, 0.f) & 0xffffu;
    const float r2 = r1 - __uint_as_float(h1 << 16);
    p0 = (uint16_t)h0;
    p1 = (uint16_t)h1;
    p2 = (uint16_t)(pack_bf16_rne(r2, 0.f) & 0xffffu);
}
// ---- precision mode f16x2: fp32 values as TWO fp16 terms of x S (S a power of two): h0 = fp16(x S), h1 = fp16(x S - h0); the
// subtraction is exact and h0 + h1 carries 22 significant bits of x S (fp16 keeps its subnormals on gfx950, in the conversion and in
// the MFMA -- tools/mfma_power_probe.hip -- so below |x S| = 2^-3 the error is absolute, <= 2^-25).  A product keeps a0 b0 + a0 b1 +
// a1 b0: three MFMA products instead of bf16x3's six, at an error of ~2^-22 per product, which after K >= 64 products is below what
// the fp32 accumulation itself commits (gemm_split_sw.hip).  |x S| > 65504 saturates; the producers report it (`ovf`).
constexpr float F16X2_ACT_SCALE = 16.0f;       // activations: full precision for 2^-7 <= |x| < 4094
constexpr float F16X2_MAX = 65504.0f;
using h2_t = __attribute__((ext_vector_type(2))) _Float16;
__device__ __forceinline__ unsigned pack_f16_rne(float lo, float hi) {
    const h2_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split2h_pack4(const f32x4_t& x, float s, u32x2_t& p0, u32x2_t& p1, bool& ovf) {
    f32x4_t y;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float t = x[i] * s;
        ovf |= !(fabsf(t) <= F16X2_MAX);
        y[i] = __builtin_amdgcn_fmed3f(t, -F16X2_MAX, F16X2_MAX);
    }
    const h2_t a = {(_Float16)y[0], (_Float16)y[1]}, b = {(_Float16)y[2], (_Float16)y[3]};
    p0[0] = __builtin_bit_cast(unsigned, a);
    p0[1] = __builtin_bit_cast(unsigned, b);
    p1[0] = pack_f16_rne(y[0] - (float)a[0], y[1] - (float)a[1]);
    p1[1] = pack_f16_rne(y[2] - (float)b[0], y[3] - (float)b[1]);
}
__device__ __forceinline__ void split2h_one(float x, float s, uint16_t& p0, uint16_t& p1, bool& ovf) {
    const float t = x * s;
    ovf |= !(fabsf(t) <= F16X2_MAX);
    const float y = __builtin_amdgcn_fmed3f(t, -F16X2_MAX, F16X2_MAX);
    const _Float16 h = (_Float16)y;
    p0 = __builtin_bit_cast(uint16_t, h);
    p1 = __builtin_bit_cast(uint16_t, (_Float16)(y - (float)h));
}
// four values -> their planes at p, p + plane, (p + 2 plane); `ovf` only moves in the fp16 format
__device__ __forceinline__ void store_planes4(uint16_t* __restrict__ p, int64_t plane, int fmt, const f32x4_t& v, bool& ovf) {
    u32x2_t p0, p1, p2;
    if (fmt == PF_F16X2) {
        split2h_pack4(v, F16X2_ACT_SCALE, p0, p1, ovf);
        *reinterpret_cast<u32x2_t*>(p) = p0;
        *reinterpret_cast<u32x2_t*>(p + plane) = p1;
    } else {
        split3_pack4(v, p0, p1, p2);
        *reinterpret_cast<u32x2_t*>(p) = p0;
        *reinterpret_cast<u32x2_t*>(p + plane) = p1;
        *reinterpret_cast<u32x2_t*>(p + 2 * plane) = p2;
    }
}
// a producer's end: lanes that saw a saturated value set the sticky flag (range_flag: int in device memory, may be null)
__device__ __forceinline__ void report_overflow(int* range_flag, bool ovf) {
    if (range_flag && ovf) atomicOr(range_flag, 1);      // (exceptional: no need to elect one lane)
}
// wave64 all-reduce sum via DPP/shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
#endif

}  // namespace w2v2
