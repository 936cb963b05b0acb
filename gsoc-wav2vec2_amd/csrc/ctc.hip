// CTC negative log-likelihood and its gradient w.r.t. the logits, fp64 recursions.
//
// Reference: CTCLoss.call (losses.py:14-45) -> tf.nn.ctc_loss(labels, logits,
// label_length, logit_length, logits_time_major=False, blank_index=pad_id).
// Per sample: log-softmax over the vocabulary, the alpha recursion over the
// blank-interleaved label string (2U+1 states), nll = -log p(labels | logits).
// Gradient (what TF's registered gradient returns for the unnormalised logits):
//   d nll / d logits[t, v] = softmax(logits[t])[v] - sum_{s: ext[s]=v} alpha_t(s) beta_t(s) / (y_t(v) p)
//
// One workgroup per sample; the recursion is sequential in t, parallel over the
// <= 513 states.  The loss sums ~768 log-probabilities of magnitude ~1 into a
// value of magnitude ~1e3, so the recursions run in fp64 (ulp(1e3) in fp32 is
// 6e-5 per step); fp64 VALU is plentiful on gfx950 and this stage is latency-
// bound on the per-frame barrier, not on arithmetic.
#include <mutex>

#include "common.h"

namespace w2v2 {
namespace {

constexpr int CTC_THREADS = 576;     // >= 2 U + 1 = 513 extended states at U = 256: one state per thread and recursion step (256: 0.77 ms at T = 768, B = 32)
constexpr double NEG_INF = -1e300;   // finite sentinel: keeps (a - m) well-defined

// log(sum exp) of two / three fp64 states.  The state values and the running sums stay fp64 (the loss adds ~768
// log-probabilities into a value of ~1e3); the transcendental parts work on the DIFFERENCES to the maximum, which lie
// in [-inf, 0] and need no more than fp32: exp and log of fp32 cost ~12 instructions each against ~90 for the fp64
// library routines, and the recursion is exactly that arithmetic, 2 x T times per sample (3.3 ms -> 0.9 ms at T = 768).
// Per-step error ~1e-7 absolute, unbiased (libm expf / logf, not the v_exp / v_log approximations).
__device__ __forceinline__ double lse2(double a, double b) {
    const double m = a > b ? a : b;
    if (m <= NEG_INF) return NEG_INF;
    return m + (double)logf(expf((float)(a - m)) + expf((float)(b - m)));
}
__device__ __forceinline__ double lse3(double a, double b, double c) {
    double m = a > b ? a : b;
    m = m > c ? m : c;
    if (m <= NEG_INF) return NEG_INF;
    return m + (double)logf(expf((float)(a - m)) + expf((float)(b - m)) + expf((float)(c - m)));
}

struct CtcArgs {
    const float* logits;      // (B, T, V)
    const int32_t* labels;    // (B, U)
    const int32_t* label_len; // (B), or null: the reference's rule, count of labels != blank (losses.py:32-33)
    const int32_t* logit_len; // (B), or null: every row takes uniform_len frames (losses.py:29-30)
    int uniform_len;
    float grad_div;           // the gradient is divided by it (division_factor, losses.py:45: `loss / division_factor`; TF's RealDiv gradient is g / y)
    float* nll;               // (B)
    float* grad;              // (B, T, V) or null
    double* alpha_ws;         // (B, T, S_max) when grad != null
    double* beta_ws;          // (B, T, S_max) when grad != null
    double* lse_ws;           // (B, T)        when grad != null
    int B, T, V, U, blank, S_max, CH;
};

// The recursions are a dependent chain of T steps per sample, so anything with memory latency inside a step is paid T
// times: the first version read its emission log-probabilities (and, in the backward sweep, alpha) from global memory
// every step and spent 2 us per step doing it (3.3 ms at T = 768).  Now a sweep touches only LDS and issues
// fire-and-forget stores: the sample's logits are staged in LDS in chunks of CH frames (the whole utterance when
// T <= 768), alpha and beta go out to a workspace, and the gradient -- independent across frames -- is a separate,
// fully parallel kernel.
//
// dynamic LDS: double lse[T]; double ab[2][S_max]; int ext[S_max]; float lg[CH][V]
__global__ __launch_bounds__(CTC_THREADS) void ctc_kernel(CtcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    double* lse = reinterpret_cast<double*>(raw);
    double* buf0 = lse + a.T;
    double* buf1 = buf0 + a.S_max;
    int* ext = reinterpret_cast<int*>(buf1 + a.S_max);
    float* lgs = reinterpret_cast<float*>(ext + ((a.S_max + 3) & ~3));
    __shared__ double nll_sh;
    __shared__ int bad_label;

    const int b = blockIdx.x, tid = threadIdx.x;
    int U;
    if (a.label_len) {
        U = a.label_len[b];
    } else {                                             // count of non-blank labels (block-wide, integer: order-independent)
        __shared__ int cnt_sh;
        if (tid == 0) cnt_sh = 0;
        __syncthreads();
        int c = 0;
        for (int u = tid; u < a.U; u += CTC_THREADS) c += a.labels[(int64_t)b * a.U + u] != a.blank;
        if (c) atomicAdd(&cnt_sh, c);
        __syncthreads();
        U = cnt_sh;
    }
    U = U < 0 ? 0 : (U > a.U ? a.U : U);
    int Tb = a.logit_len ? a.logit_len[b] : a.uniform_len;
    Tb = Tb < 0 ? 0 : (Tb > a.T ? a.T : Tb);
    const int S = 2 * U + 1;
    const float* __restrict__ lg = a.logits + (int64_t)b * a.T * a.V;

    // A label outside [0, V) (a vocabulary / config mismatch, a -1 pad) would index the staged logits out of bounds:
    // the sample's loss becomes NaN instead (its gradient rows are zeros), and the state uses the blank in its place.
    if (tid == 0) bad_label = 0;
    __syncthreads();
    for (int s = tid; s < S; s += CTC_THREADS) {
        int e = a.blank;
        if (s & 1) {
            e = a.labels[(int64_t)b * a.U + (s >> 1)];
            if (e < 0 || e >= a.V) { bad_label = 1; e = a.blank; }
        }
        ext[s] = e;
    }
    // log-sum-exp per frame (fp64)
    for (int t = tid; t < Tb; t += CTC_THREADS) {
        const float* r = lg + (int64_t)t * a.V;
        float m = r[0];
        for (int v = 1; v < a.V; ++v) m = fmaxf(m, r[v]);
        double acc = 0.0;
        for (int v = 0; v < a.V; ++v) acc += exp((double)r[v] - (double)m);
        lse[t] = (double)m + log(acc);
        if (a.lse_ws) a.lse_ws[(int64_t)b * a.T + t] = lse[t];
    }
    __syncthreads();
    // With a gradient the two sweeps are independent until the gradient kernel: grid.y = 2 runs alpha (and the loss) in
    // block (b, 0) and beta in block (b, 1) side by side -- the kernel is a latency-bound recursion over T, one block
    // per sample, so this halves its time (1.47 -> 0.75 ms at B = 32, T = 768, U = 256).
    const bool do_alpha = blockIdx.y == 0, do_beta = a.grad && (gridDim.y == 1 || blockIdx.y == 1);
    if (Tb == 0) {
        if (tid == 0 && do_alpha) a.nll[b] = U == 0 ? 0.0f : INFINITY;
        return;                                   // (the gradient kernel writes zeros for this sample)
    }
    // stage the frames of chunk c = [c CH, (c + 1) CH) of this sample's logits in LDS (coalesced)
    int chunk = -1;
    auto stage = [&](int c) {
        __syncthreads();                          // everyone is done with the previous chunk
        const int f0 = c * a.CH, nf = min(a.CH, Tb - f0);
        const float* src = lg + (int64_t)f0 * a.V;
        for (int i = tid; i < nf * a.V; i += CTC_THREADS) lgs[i] = src[i];
        chunk = c;
        __syncthreads();
    };
    auto logp = [&](int t, int s) { return (double)lgs[(t - chunk * a.CH) * a.V + ext[s]] - lse[t]; };

    // ---- alpha ----
    if (do_alpha) {
    double* prev = buf0;
    double* cur = buf1;
    double* aw = a.grad ? a.alpha_ws + (int64_t)b * a.T * a.S_max : nullptr;
    stage(0);
    for (int s = tid; s < S; s += CTC_THREADS) {
        const double v = s < 2 ? logp(0, s) : NEG_INF;
        prev[s] = v;
        if (aw) aw[s] = v;
    }
    __syncthreads();
    for (int t = 1; t < Tb; ++t) {
        if (t / a.CH != chunk) stage(t / a.CH);   // block-uniform
        for (int s = tid; s < S; s += CTC_THREADS) {
            const double a0 = prev[s];
            const double a1 = s >= 1 ? prev[s - 1] : NEG_INF;
            const bool skip = s >= 2 && (s & 1) && ext[s] != ext[s - 2];
            const double a2 = skip ? prev[s - 2] : NEG_INF;
            double v = lse3(a0, a1, a2);
            v = v <= NEG_INF ? NEG_INF : v + logp(t, s);
            cur[s] = v;
            if (aw) aw[(int64_t)t * a.S_max + s] = v;
        }
        __syncthreads();
        double* tmp = prev; prev = cur; cur = tmp;
    }
    if (tid == 0) {
        const double tot = S >= 2 ? lse2(prev[S - 1], prev[S - 2]) : prev[0];
        nll_sh = tot <= NEG_INF ? (double)INFINITY : -tot;
        a.nll[b] = bad_label ? __builtin_nanf("") : (float)nll_sh;
    }
    __syncthreads();
    }
    if (!do_beta) return;

    // ---- beta (stored; the gradient kernel combines it with alpha) ----
    double* bw = a.beta_ws + (int64_t)b * a.T * a.S_max;
    double* bprev = buf0;
    double* bcur = buf1;
    for (int t = Tb - 1; t >= 0; --t) {
        if (t / a.CH != chunk) stage(t / a.CH);
        for (int s = tid; s < S; s += CTC_THREADS) {
            double v;
            if (t == Tb - 1) {
                v = (s >= S - 2) ? logp(t, s) : NEG_INF;
            } else {
                const double b0 = bprev[s];
                const double b1 = s + 1 < S ? bprev[s + 1] : NEG_INF;
                const bool skip = s + 2 < S && (s & 1) && ext[s] != ext[s + 2];
                const double b2 = skip ? bprev[s + 2] : NEG_INF;
                v = lse3(b0, b1, b2);
                v = v <= NEG_INF ? NEG_INF : v + logp(t, s);
            }
            bcur[s] = v;
            bw[(int64_t)t * a.S_max + s] = v;
        }
        __syncthreads();
        double* tmp = bprev; bprev = bcur; bcur = tmp;
    }
}

// d nll / d logits[t, v] = softmax(logits[t])[v] - sum_{s: ext[s] = v} alpha_t(s) beta_t(s) / (y_t(v) p): one block per
// (sample, frame), no dependence between frames.
__global__ __launch_bounds__(64) void ctc_grad_kernel(CtcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    // occupancy per vocabulary entry, accumulated in 2^-61 fixed point with INTEGER atomics: the sum does not depend on the
    // order the lanes arrive in, so the gradient is bitwise reproducible (fp64 atomicAdd was not); every term is a
    // probability in [0, 1] and the terms of one entry sum to at most 1, so 2^61 leaves headroom in 64 bits
    unsigned long long* occ = reinterpret_cast<unsigned long long*>(raw);          // [V]
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    int U;
    if (a.label_len) {
        U = a.label_len[b];
    } else {
        int c = 0;
        for (int u = tid; u < a.U; u += 64) c += a.labels[(int64_t)b * a.U + u] != a.blank;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        U = c;
    }
    U = U < 0 ? 0 : (U > a.U ? a.U : U);
    int Tb = a.logit_len ? a.logit_len[b] : a.uniform_len;
    Tb = Tb < 0 ? 0 : (Tb > a.T ? a.T : Tb);
    const int S = 2 * U + 1;
    float* __restrict__ gr = a.grad + ((int64_t)b * a.T + t) * a.V;
    if (t >= Tb || !isfinite(a.nll[b])) {               // frames past logit_length, or an infeasible alignment
        for (int v = tid; v < a.V; v += 64) gr[v] = 0.f;
        return;
    }
    for (int v = tid; v < a.V; v += 64) occ[v] = 0ull;
    __syncthreads();
    const float* __restrict__ lg = a.logits + ((int64_t)b * a.T + t) * a.V;
    const double lse = a.lse_ws[(int64_t)b * a.T + t];
    const double* __restrict__ aw = a.alpha_ws + ((int64_t)b * a.T + t) * a.S_max;
    const double* __restrict__ bw = a.beta_ws + ((int64_t)b * a.T + t) * a.S_max;
    // the float nll would cost 1e-4 relative in every weight: recompute it in fp64 from the two sweeps at this frame
    //   p = sum_s alpha_t(s) beta_t(s) / y_t(ext[s])   (any t)
    auto label_at = [&](int s) {
        if (!(s & 1)) return a.blank;
        const int e = a.labels[(int64_t)b * a.U + (s >> 1)];
        return (e < 0 || e >= a.V) ? a.blank : e;        // (such a sample has a NaN loss and takes the zero-gradient exit above)
    };
    double m = NEG_INF;
    for (int s = tid; s < S; s += 64) {
        const int e = label_at(s);
        const double w = (aw[s] > NEG_INF && bw[s] > NEG_INF) ? aw[s] + bw[s] - ((double)lg[e] - lse) : NEG_INF;
        m = w > m ? w : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    double tot = 0.0;
    for (int s = tid; s < S; s += 64) {
        const int e = label_at(s);
        if (aw[s] > NEG_INF && bw[s] > NEG_INF) tot += exp(aw[s] + bw[s] - ((double)lg[e] - lse) - m);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
    const double logp_total = m + log(tot);              // = -nll in fp64
    const double FIX = 2305843009213693952.0;            // 2^61
    for (int s = tid; s < S; s += 64) {
        const int e = label_at(s);
        if (aw[s] > NEG_INF && bw[s] > NEG_INF) {
            const double c = exp(aw[s] + bw[s] - ((double)lg[e] - lse) - logp_total);
            atomicAdd(&occ[e], (unsigned long long)__double2ull_rn(fmin(c, 1.0) * FIX));
        }
    }
    __syncthreads();
    // (/ grad_div as a separate fp32 division: the same bits as dividing the fp32 gradient afterwards, and exact for 1.0)
    for (int v = tid; v < a.V; v += 64) gr[v] = (float)(exp((double)lg[v] - lse) - (double)occ[v] * (1.0 / FIX)) / a.grad_div;
}

// loss_sum[0] = sum_b nll[b] / div, added in row order by one lane (Keras Reduction.SUM of the per-sample losses / division_factor)
__global__ void ctc_loss_sum_kernel(const float* __restrict__ nll, int B, float div, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += nll[b] / div;
        out[0] = s;
    }
}

struct LenArgs {
    int32_t ks[W2V2_MAX_CONV_LAYERS], ss[W2V2_MAX_CONV_LAYERS];
    int nl;
};

// attention mask (B, L) of 0 / 1 -> valid frames per row.  Stage 1: (chunks, B) blocks sum 16-byte pieces of a row and add
// their count to out[b] (integer atomics: order-independent); stage 2 turns the B sums into frame counts.  (The first
// version gave each row ONE block: 0.8 ms for 16 x 480000 samples, a latency-bound crawl over 1.9 MB per block.)
__global__ __launch_bounds__(256) void mask_sum_kernel(const int32_t* __restrict__ mask, int32_t* __restrict__ out, int64_t L, int vec) {
    __shared__ int red[4];
    const int b = blockIdx.y;
    const int32_t* row = mask + (int64_t)b * L;
    int acc = 0;
    if (vec) {
        const int64_t n4 = L >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const int4 v = reinterpret_cast<const int4*>(row)[i];
            acc += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < L; i += (int64_t)gridDim.x * 256) acc += row[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + b, (red[0] + red[1]) + (red[2] + red[3]));
}

__global__ void frame_len_finish_kernel(int32_t* __restrict__ out, int B, LenArgs la) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    long long n = out[b];
    // 1 + (len - k) // s with floor semantics (modeling.py:202-204)
    for (int i = 0; i < la.nl; ++i) {
        const long long d = n - la.ks[i];
        const long long q = d >= 0 ? d / la.ss[i] : -((-d + la.ss[i] - 1) / la.ss[i]);
        n = 1 + q;
    }
    out[b] = (int32_t)(n < 0 ? 0 : n);
}


}  // namespace

int launch_ctc(Profiler* prof, const float* logits, int B, int T, int V, const int32_t* labels,
               int U, const int32_t* label_len, const int32_t* logit_len, int blank, float* nll,
               float* grad, hipStream_t s) {
    W2V2_REQUIRE(label_len && logit_len, "ctc: null operand");
    return launch_ctc_x(prof, logits, B, T, V, labels, U, label_len, logit_len, 0, blank, 1.0f, nll, grad, nullptr, s);
}

int launch_ctc_x(Profiler* prof, const float* logits, int B, int T, int V, const int32_t* labels, int U, const int32_t* label_len,
                 const int32_t* logit_len, int uniform_len, int blank, float grad_div, float* nll, float* grad, float* loss_sum, hipStream_t s) {
    W2V2_REQUIRE(logits && labels && nll && (logit_len || uniform_len > 0), "ctc: null operand");
    W2V2_REQUIRE(B > 0 && T > 0 && V > 0 && U >= 0, "ctc: bad sizes");
    W2V2_REQUIRE(blank >= 0 && blank < V, "ctc: blank index %d outside vocabulary %d", blank, V);
    CtcArgs a;
    a.logits = logits; a.labels = labels; a.label_len = label_len; a.logit_len = logit_len;
    a.uniform_len = uniform_len; a.grad_div = grad_div;
    a.nll = nll; a.grad = grad; a.B = B; a.T = T; a.V = V; a.U = U; a.blank = blank;
    a.S_max = 2 * U + 1;
    a.alpha_ws = a.beta_ws = a.lse_ws = nullptr;
    if (grad) {
        const size_t per = (size_t)B * T * a.S_max;
        const size_t need = (2 * per + (size_t)B * T) * sizeof(double);
        void* raw = nullptr;                    // alpha | beta | lse, fp64: per-stream scratch owned by the library
        if (int e = stream_scratch(SCRATCH_CTC, s, need, &raw)) return e;
        double* ws = reinterpret_cast<double*>(raw);
        a.alpha_ws = ws;
        a.beta_ws = ws + per;
        a.lse_ws = ws + 2 * per;
    }
    // logits chunk in LDS: the whole utterance if it fits next to the state arrays, else as many frames as do
    const size_t fixed = (size_t)(T + 2 * a.S_max) * sizeof(double) + (size_t)((a.S_max + 3) & ~3) * sizeof(int) + 16;
    W2V2_REQUIRE(fixed + (size_t)V * sizeof(float) <= 150 * 1024, "ctc: T=%d U=%d needs %zu B of LDS", T, U, fixed);
    size_t frames = (150 * 1024 - fixed) / ((size_t)V * sizeof(float));
    a.CH = (int)(frames < (size_t)T ? frames : (size_t)T);
    const size_t lds = fixed + (size_t)a.CH * V * sizeof(float);
    static std::once_flag attr_once;                       // (several host threads may each drive their own model)
    hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    });
    W2V2_HIP_CHECK(attr_err);
    ProfScope ps(prof, FAM_CTC, 30.0 * B * (double)T * a.S_max, 4.0 * B * (double)T * V * (grad ? 2 : 1), s);
    W2V2_LAUNCH(ctc_kernel, dim3(B, grad ? 2 : 1), dim3(CTC_THREADS), lds, s, a);
    if (grad) W2V2_LAUNCH(ctc_grad_kernel, dim3(T, B), dim3(64), (size_t)V * sizeof(unsigned long long), s, a);
    if (loss_sum) W2V2_LAUNCH(ctc_loss_sum_kernel, dim3(1), dim3(64), 0, s, nll, B, grad_div, loss_sum);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

int launch_frame_lengths(Profiler* prof, const int32_t* mask, int32_t* frame_len, int B, int64_t L,
                         const int32_t* ks, const int32_t* ss, int nl, hipStream_t s) {
    W2V2_REQUIRE(mask && frame_len && ks && ss, "frame_lengths: null operand");
    W2V2_REQUIRE(B > 0 && L > 0 && nl > 0 && nl <= W2V2_MAX_CONV_LAYERS, "frame_lengths: bad sizes");
    ProfScope ps(prof, FAM_MISC, 0.0, 4.0 * B * (double)L, s);
    LenArgs la;
    la.nl = nl;
    for (int i = 0; i < nl; ++i) {
        W2V2_REQUIRE(ss[i] > 0, "frame_lengths: stride must be positive");
        la.ks[i] = ks[i];
        la.ss[i] = ss[i];
    }
    W2V2_HIP_CHECK(hipMemsetAsync(frame_len, 0, (size_t)B * sizeof(int32_t), s));
    const int vec = (L % 4 == 0) && (reinterpret_cast<uintptr_t>(mask) & 15) == 0;
    int64_t chunks = (L + 8191) / 8192;                       // >= 32 loads per lane before another block is worth it
    chunks = chunks < 1 ? 1 : (chunks > 128 ? 128 : chunks);
    W2V2_LAUNCH(mask_sum_kernel, dim3((unsigned)chunks, B), dim3(256), 0, s, mask, frame_len, L, vec);
    W2V2_LAUNCH(frame_len_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, s, frame_len, B, la);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace w2v2
