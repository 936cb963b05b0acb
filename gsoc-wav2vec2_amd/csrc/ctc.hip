// CTC negative log-likelihood and its gradient w.r.t. the logits, fp64 recursions in the PROBABILITY domain.
//
// Reference: CTCLoss.call (losses.py:14-45) -> tf.nn.ctc_loss(labels, logits,
// label_length, logit_length, logits_time_major=False, blank_index=pad_id).
// Per sample: softmax over the vocabulary, the alpha recursion over the
// blank-interleaved label string (2U+1 states), nll = -log p(labels | logits).
// Gradient (what TF's registered gradient returns for the unnormalised logits):
//   d nll / d logits[t, v] = softmax(logits[t])[v] - sum_{s: ext[s]=v} alpha_t(s) beta_t(s) / (y_t(v) p)
//
// One workgroup per sample and sweep; the recursion is sequential in t, so what a step costs is paid T times (768 / 1499).
// Rounds 1-5 ran it in log space: one thread per state, log-sum-exp of three fp64 states per step (three expf + one logf on the
// differences), nine waves at the per-frame barrier -- 0.73 us per step, 0.56 ms per fine-tune step at T = 768 and 1.04 ms at
// T = 1499, all of it latency.  Round 6: the same recursion on PROBABILITIES with a separate binary exponent,
//   alpha_t(s) = (alpha_{t-1}(s) + alpha_{t-1}(s-1) + [skip] alpha_{t-1}(s-2)) y_t(ext[s]),
// two adds and a multiply per state and no transcendental inside the sweep: y_t = softmax(logits[t]) comes from a separate,
// fully parallel kernel (one wave per frame, fp64), and a thread loads the two emissions it needs -- y_t(blank), y_t(label k) --
// CTC_PF steps ahead, so their L2 latency never sits on the recursion (and no frame count is bounded by LDS).  A thread owns the state
// PAIR (2k: blank, 2k+1: label k) in registers as two fp64 mantissas with ONE int32 exponent E_k; every CTC_NORM steps the pair is
// renormalised so that the larger mantissa lies in [1, 2) (exponent-field reads and v_ldexp_f64: a handful of instructions, off
// the other steps' dependency chain).  The only values that cross threads per step are the neighbour's odd state (alpha) or both
// states (beta) and its exponent, through a double-buffered LDS array and ONE barrier among five waves; the receiver brings them to
// its own exponent with one ldexp (or adopts the neighbour's, if that is 2^600 above its own: what it held is then below any
// rounding).  A state 2^-1022 below its pair partner or below the mass that flows into the pair is dropped, exactly as its weight
// deserves, and nothing else under- or overflows: between two renormalisations a mantissa shrinks by at most the product of
// CTC_NORM emission probabilities and grows by at most 3^CTC_NORM, and the dynamic range ACROSS states is that of the int32
// exponent, i.e. the robustness of log space (a blank-dominated model early in training has alpha(all-blank prefix) /
// alpha(best path) far beyond 1e308) at the cost of fixed point.  Sums and products are fp64 (relative error 1e-16 per step,
// against ~1e-7 per step of the fp32 exp / log on differences above).  Inputs that break it: a frame in which a needed label has
// softmax probability below e^-745 (fp64 exp flushes to 0), or CTC_NORM consecutive frames whose needed labels all sit below
// 1e-77: that path is dropped where log space would have charged it its > 700 nats.
#include "common.h"

namespace w2v2 {
namespace {

constexpr int CTC_THREADS = 320;     // >= U + 1 = 257 state pairs at U = 256; five waves at the per-step barrier
constexpr int CTC_PF = 16;           // steps the two emission probabilities of a step are loaded ahead of it
constexpr int CTC_NORM = 4;          // steps between two renormalisations of a pair's mantissas
constexpr int CTC_NOEXP = -(1 << 28);    // exponent of a pair whose states are both zero: loses every max()
constexpr int CTC_SHIFT_MIN = -2200;     // (ldexp by less than this is zero for every normalised mantissa: clamp, the difference may be ~ -2^28)

struct CtcArgs {
    const float* logits;      // (B, T, V)
    const int32_t* labels;    // (B, U)
    const int32_t* label_len; // (B), or null: the reference's rule, count of labels != blank (losses.py:32-33)
    const int32_t* logit_len; // (B), or null: every row takes uniform_len frames (losses.py:29-30)
    int uniform_len;
    float grad_div;           // the gradient is divided by it (division_factor, losses.py:45: `loss / division_factor`; TF's RealDiv gradient is g / y)
    float* nll;               // (B)
    float* grad;              // (B, T, V) or null
    double* alpha_ws;         // (B, T, SP) when grad != null: mantissas of alpha_t(s), SP = 2 (U + 1); the exponent of states 2k, 2k+1 is alpha_ex[.., k]
    double* beta_ws;          // (B, T, SP) when grad != null: beta likewise
    int32_t* alpha_ex;        // (B, T, U + 1)
    int32_t* beta_ex;         // (B, T, U + 1)
    double* y_ws;             // (B, T, V): softmax(logits), written by ctc_softmax_kernel
    int B, T, V, U, blank, SP;
};

// Barrier of the recursion's steps: what must be visible across the block are the LDS exchange entries, so the wave waits for its LDS
// traffic only.  __syncthreads() also drains vmcnt -- the step's fire-and-forget stores of alpha / beta to the workspace -- and a
// store's round trip (~1 us) then becomes the step time (rounds 1-5 paid exactly that: 0.73 us per step).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int f64_exponent(double v) { return (int)((__double_as_longlong(v) >> 52) & 0x7FF); }
__device__ __forceinline__ double shift2(double v, int sh) { return ldexp(v, sh < CTC_SHIFT_MIN ? CTC_SHIFT_MIN : sh); }
// bring the pair (ev, od) x 2^E to the form where the larger mantissa lies in [1, 2); both zero (or subnormal): E = CTC_NOEXP
__device__ __forceinline__ void normalise_pair(double& ev, double& od, int& E) {
    const int ex = max(f64_exponent(ev), f64_exponent(od));
    if (ex == 0) {
        ev = od = 0.0;
        E = CTC_NOEXP;
    } else {
        const int sh = 1023 - ex;
        ev = ldexp(ev, sh);
        od = ldexp(od, sh);
        E -= sh;
    }
}

// y = softmax(logits[b, t, :]) in fp64, one wave per frame (lanes stride over the vocabulary; fixed butterflies: one summation order)
__global__ __launch_bounds__(256) void ctc_softmax_kernel(CtcArgs a) {
    const int64_t frame = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= (int64_t)a.B * a.T) return;
    const int lane = threadIdx.x & 63;
    const float* __restrict__ r = a.logits + frame * a.V;
    double* __restrict__ y = a.y_ws + frame * a.V;
    float m = -INFINITY;
    for (int v = lane; v < a.V; v += 64) m = fmaxf(m, r[v]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    double acc = 0.0;
    for (int v = lane; v < a.V; v += 64) {
        const double e = exp((double)r[v] - (double)m);
        y[v] = e;
        acc += e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    const double inv = 1.0 / acc;
    for (int v = lane; v < a.V; v += 64) y[v] *= inv;      // (each lane rescales what it wrote itself)
}

// dynamic LDS: double xv[2][2 (U + 3)]; int xe[2][U + 3]; int lab[U + 1]
__global__ __launch_bounds__(CTC_THREADS) void ctc_kernel(CtcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    const int XS = 2 * (a.U + 3), ES = a.U + 3;          // doubles / ints per exchange buffer
    double* xv = reinterpret_cast<double*>(raw);         // [2][XS]
    int* xe = reinterpret_cast<int*>(xv + 2 * XS);       // [2][ES]
    int* lab = xe + ((2 * ES + 3) & ~3);                 // [U + 1]
    __shared__ int bad_label, cnt_sh;

    const int b = blockIdx.x, tid = threadIdx.x;
    int U;
    if (tid == 0) { cnt_sh = 0; bad_label = 0; }
    __syncthreads();
    if (a.label_len) {
        U = a.label_len[b];
    } else {                                             // count of non-blank labels (block-wide, integer: order-independent)
        int c = 0;
        for (int u = tid; u < a.U; u += CTC_THREADS) c += a.labels[(int64_t)b * a.U + u] != a.blank;
        if (c) atomicAdd(&cnt_sh, c);
        __syncthreads();
        U = cnt_sh;
    }
    U = U < 0 ? 0 : (U > a.U ? a.U : U);
    int Tb = a.logit_len ? a.logit_len[b] : a.uniform_len;
    Tb = Tb < 0 ? 0 : (Tb > a.T ? a.T : Tb);
    // A label outside [0, V) (a vocabulary / config mismatch, a -1 pad) would index the frame's probabilities out of bounds:
    // the sample's loss becomes NaN instead (its gradient rows are zeros), and the state uses the blank in its place.
    for (int u = tid; u <= a.U; u += CTC_THREADS) {
        int e = a.blank;
        if (u < U) {
            e = a.labels[(int64_t)b * a.U + u];
            if (e < 0 || e >= a.V) { bad_label = 1; e = a.blank; }
        }
        lab[u] = e;
    }
    for (int i = tid; i < 2 * XS; i += CTC_THREADS) xv[i] = 0.0;       // (the guard entries stay "no such state": zero, no exponent)
    for (int i = tid; i < 2 * ES; i += CTC_THREADS) xe[i] = CTC_NOEXP;
    __syncthreads();
    // With a gradient the two sweeps are independent until the gradient kernel: grid.y = 2 runs alpha (and the loss) in
    // block (b, 0) and beta in block (b, 1) side by side.
    const bool do_alpha = blockIdx.y == 0, do_beta = a.grad && (gridDim.y == 1 || blockIdx.y == 1);
    if (Tb == 0) {
        if (tid == 0 && do_alpha) a.nll[b] = U == 0 ? 0.0f : INFINITY;
        return;                                   // (the gradient kernel writes zeros for this sample)
    }
    const double* __restrict__ yg = a.y_ws + (int64_t)b * a.T * a.V;      // this sample's softmax rows
    // the neighbour's value at this thread's exponent; if the neighbour sits 2^600 above, the thread adopts ITS exponent (own states: below any rounding)
    auto align = [&](int Enb, double& ev, double& od, int& E) -> int {
        int d = Enb - E;
        if (d > 600) {
            ev = od = 0.0;
            E = Enb;
            d = 0;
        }
        return d < CTC_SHIFT_MIN ? CTC_SHIFT_MIN : d;
    };

    const int k = tid;                            // this thread's state pair: 2k (blank), 2k + 1 (label k)
    const bool in_range = k <= a.U, has_even = k <= U, has_odd = k < U;
    const int my_lab = in_range ? lab[k] : a.blank;

    // ---- alpha ----
    if (do_alpha) {
        const bool skip = has_odd && k >= 1 && my_lab != lab[k - 1];      // state 2k+1 may be entered from 2k-1
        double* aw = a.grad ? a.alpha_ws + (int64_t)b * a.T * a.SP : nullptr;
        int32_t* ae = a.grad ? a.alpha_ex + (int64_t)b * a.T * (a.U + 1) : nullptr;
        double ev = 0.0, od = 0.0;                // mantissas of alpha_t(2k), alpha_t(2k+1)
        int E = 0;
        if (k == 0) {
            ev = yg[a.blank];
            od = has_odd ? yg[my_lab] : 0.0;
        }
        normalise_pair(ev, od, E);
        // exchange: entry k + 1 of a buffer = thread k's odd state and exponent; entry 0: "state -1"
        auto publish = [&](int t) {
            if (in_range) {
                xv[(t & 1) * XS + k + 1] = od;
                xe[(t & 1) * ES + k + 1] = od != 0.0 ? E : CTC_NOEXP;
            }
            if (aw && has_even) {
                *reinterpret_cast<double2*>(aw + (int64_t)t * a.SP + 2 * k) = double2{ev, od};
                ae[(int64_t)t * (a.U + 1) + k] = (ev != 0.0 || od != 0.0) ? E : CTC_NOEXP;
            }
        };
        publish(0);
        double qb[CTC_PF], ql[CTC_PF];            // y_t(blank), y_t(label k) of the next CTC_PF steps
#pragma unroll
        for (int j = 0; j < CTC_PF; ++j) {
            const int tf = min(1 + j, Tb - 1);
            qb[j] = yg[(int64_t)tf * a.V + a.blank];
            ql[j] = yg[(int64_t)tf * a.V + my_lab];
        }
        auto step = [&](int t, double yb, double yl) {
            lds_barrier();
            const double pm1 = in_range ? xv[((t - 1) & 1) * XS + k] : 0.0;         // alpha_{t-1}(2k - 1) ...
            const int Ep = in_range ? xe[((t - 1) & 1) * ES + k] : CTC_NOEXP;       // ... and its exponent
            const double p = ldexp(pm1, align(Ep, ev, od, E));
            const double n_ev = has_even ? (ev + p) * yb : 0.0;
            od = has_odd ? (od + ev + (skip ? p : 0.0)) * yl : 0.0;
            ev = n_ev;
            if (t % CTC_NORM == 0) normalise_pair(ev, od, E);
            publish(t);
        };
        // whole groups of CTC_PF steps without a branch between them (the wait-count pass then counts the loads and stores in flight
        // exactly: a step waits for the emissions requested CTC_PF steps ago, not for the stores of the step before), then the tail
        int t0 = 1;
        for (; t0 + CTC_PF <= Tb; t0 += CTC_PF) {
#pragma unroll
            for (int j = 0; j < CTC_PF; ++j) {
                step(t0 + j, qb[j], ql[j]);
                const int tf = min(t0 + j + CTC_PF, Tb - 1);
                qb[j] = yg[(int64_t)tf * a.V + a.blank];
                ql[j] = yg[(int64_t)tf * a.V + my_lab];
            }
        }
#pragma unroll
        for (int j = 0; j < CTC_PF; ++j)
            if (t0 + j < Tb) step(t0 + j, qb[j], ql[j]);      // (block-uniform)
        __syncthreads();
        if (k == U) {       // total = alpha(S - 1) + alpha(S - 2) = this thread's even state + thread U - 1's odd state
            const int bufl = (Tb - 1) & 1;
            const double pm1 = xv[bufl * XS + U];
            const int Ep = xe[bufl * ES + U], En = max(E, Ep);
            const double tot = shift2(ev, E - En) + shift2(pm1, Ep - En);
            const double nll = tot > 0.0 ? -(log(tot) + (double)En * 0.69314718055994530942) : (double)INFINITY;
            a.nll[b] = bad_label ? __builtin_nanf("") : (float)nll;
        }
        __syncthreads();
    }
    if (!do_beta) return;

    // ---- beta (stored; the gradient kernel combines it with alpha).  beta_t(s) includes the emission at t, as alpha_t(s) does.
    // exchange: entries 2k, 2k + 1 of a buffer = thread k's even | odd state, exponent entry k; thread U + 1's stay "no such state"
    {
        if (do_alpha) {                            // (grid.y == 1: the alpha sweep used the buffers)
            for (int i = tid; i < 2 * XS; i += CTC_THREADS) xv[i] = 0.0;
            for (int i = tid; i < 2 * ES; i += CTC_THREADS) xe[i] = CTC_NOEXP;
            __syncthreads();
        }
        double* bw = a.beta_ws + (int64_t)b * a.T * a.SP;
        int32_t* be = a.beta_ex + (int64_t)b * a.T * (a.U + 1);
        const bool skip = has_odd && k + 1 < U && my_lab != lab[k + 1];      // state 2k+1 may move on to 2k+3
        double ev = 0.0, od = 0.0;
        int E = 0;
        auto publish = [&](int t) {
            const int Epub = (ev != 0.0 || od != 0.0) ? E : CTC_NOEXP;
            if (in_range) {
                xv[(t & 1) * XS + 2 * k] = ev;
                xv[(t & 1) * XS + 2 * k + 1] = od;
                xe[(t & 1) * ES + k] = Epub;
            }
            if (has_even) {
                *reinterpret_cast<double2*>(bw + (int64_t)t * a.SP + 2 * k) = double2{ev, od};
                be[(int64_t)t * (a.U + 1) + k] = Epub;
            }
        };
        {
            const double* yr = yg + (int64_t)(Tb - 1) * a.V;
            ev = (k == U) ? yr[a.blank] : 0.0;                          // state S - 1 = 2U
            od = (has_odd && k == U - 1) ? yr[my_lab] : 0.0;            // state S - 2 = 2U - 1
            normalise_pair(ev, od, E);
            publish(Tb - 1);
        }
        double qb[CTC_PF], ql[CTC_PF];
#pragma unroll
        for (int j = 0; j < CTC_PF; ++j) {
            const int tf = max(Tb - 2 - j, 0);
            qb[j] = yg[(int64_t)tf * a.V + a.blank];
            ql[j] = yg[(int64_t)tf * a.V + my_lab];
        }
        auto step = [&](int t, double yb, double yl) {
            lds_barrier();
            const int rb = ((t + 1) & 1);
            const double nb_ev = in_range ? xv[rb * XS + 2 * (k + 1)] : 0.0, nb_od = in_range ? xv[rb * XS + 2 * (k + 1) + 1] : 0.0;   // beta_{t+1}(2k + 2), (2k + 3)
            const int Enb = in_range ? xe[rb * ES + k + 1] : CTC_NOEXP;
            const int d = align(Enb, ev, od, E);
            const double n0 = ldexp(nb_ev, d), n1 = ldexp(nb_od, d);
            const double n_ev = has_even ? (ev + od) * yb : 0.0;
            od = has_odd ? (od + n0 + (skip ? n1 : 0.0)) * yl : 0.0;
            ev = n_ev;
            if ((Tb - 1 - t) % CTC_NORM == 0) normalise_pair(ev, od, E);
            publish(t);
        };
        int t0 = Tb - 2;
        for (; t0 - CTC_PF + 1 >= 0; t0 -= CTC_PF) {
#pragma unroll
            for (int j = 0; j < CTC_PF; ++j) {
                step(t0 - j, qb[j], ql[j]);
                const int tf = max(t0 - j - CTC_PF, 0);
                qb[j] = yg[(int64_t)tf * a.V + a.blank];
                ql[j] = yg[(int64_t)tf * a.V + my_lab];
            }
        }
#pragma unroll
        for (int j = 0; j < CTC_PF; ++j)
            if (t0 - j >= 0) step(t0 - j, qb[j], ql[j]);      // (block-uniform)
    }
}

// d nll / d logits[t, v] = softmax(logits[t])[v] - sum_{s: ext[s] = v} alpha_t(s) beta_t(s) / (y_t(v) p): one block per
// (sample, frame), no dependence between frames.  The occupancy of a state is its share of  sum_s alpha_t(s) beta_t(s) / y_t(ext[s])  (= p),
// formed from mantissas and exponents: shares are taken relative to the largest exponent of the frame.
__global__ __launch_bounds__(64) void ctc_grad_kernel(CtcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    // occupancy per vocabulary entry, accumulated in 2^-61 fixed point with INTEGER atomics: the sum does not depend on the
    // order the lanes arrive in, so the gradient is bitwise reproducible (fp64 atomicAdd was not); every term is a
    // probability in [0, 1] and the terms of one entry sum to at most 1, so 2^61 leaves headroom in 64 bits
    unsigned long long* occ = reinterpret_cast<unsigned long long*>(raw);          // [V]
    double* yv = reinterpret_cast<double*>(occ + a.V);                             // [V] softmax of this frame
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
    const double* __restrict__ yg = a.y_ws + ((int64_t)b * a.T + t) * a.V;
    for (int v = tid; v < a.V; v += 64) {
        occ[v] = 0ull;
        yv[v] = yg[v];
    }
    __syncthreads();
    const double* __restrict__ aw = a.alpha_ws + ((int64_t)b * a.T + t) * a.SP;
    const double* __restrict__ bw = a.beta_ws + ((int64_t)b * a.T + t) * a.SP;
    const int32_t* __restrict__ ae = a.alpha_ex + ((int64_t)b * a.T + t) * (a.U + 1);
    const int32_t* __restrict__ be = a.beta_ex + ((int64_t)b * a.T + t) * (a.U + 1);
    auto label_at = [&](int s) {
        if (!(s & 1)) return a.blank;
        const int e = a.labels[(int64_t)b * a.U + (s >> 1)];
        return (e < 0 || e >= a.V) ? a.blank : e;        // (such a sample has a NaN loss and takes the zero-gradient exit above)
    };
    // weight of state s = alpha beta / y as (mantissa, exponent); zero mantissa: the state is unreachable (or y flushed: then alpha is 0 too)
    auto weight = [&](int s, int e, int& ex) {
        const double ab = aw[s] * bw[s];
        ex = ae[s >> 1] + be[s >> 1];
        return ab > 0.0 ? ab / yv[e] : 0.0;
    };
    int emax = 2 * CTC_NOEXP;
    for (int s = tid; s < S; s += 64) {
        int ex;
        if (weight(s, label_at(s), ex) > 0.0) emax = max(emax, ex);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) emax = max(emax, __shfl_xor(emax, off, 64));
    double tot = 0.0;                                    // lane-strided partial sums, then a fixed butterfly: one order, every run
    for (int s = tid; s < S; s += 64) {
        int ex;
        const double w = weight(s, label_at(s), ex);
        if (w > 0.0) tot += shift2(w, ex - emax);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
    const double FIX = 2305843009213693952.0;            // 2^61
    if (tot > 0.0 && isfinite(tot)) {
        const double itot = 1.0 / tot;
        for (int s = tid; s < S; s += 64) {
            const int e = label_at(s);
            int ex;
            const double w = weight(s, e, ex);
            const double c = w > 0.0 ? shift2(w, ex - emax) * itot : 0.0;
            if (c > 0.0) atomicAdd(&occ[e], (unsigned long long)__double2ull_rn(fmin(c, 1.0) * FIX));
        }
    }
    __syncthreads();
    // (/ grad_div as a separate fp32 division: the same bits as dividing the fp32 gradient afterwards, and exact for 1.0)
    for (int v = tid; v < a.V; v += 64) gr[v] = (float)(yv[v] - (double)occ[v] * (1.0 / FIX)) / a.grad_div;
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
    a.SP = 2 * (U + 1);
    a.alpha_ws = a.beta_ws = nullptr;
    a.alpha_ex = a.beta_ex = nullptr;
    {
        const size_t per = grad ? (size_t)B * T * a.SP : 0, per_ex = grad ? (((size_t)B * T * (U + 1) + 3) & ~(size_t)3) : 0;
        const size_t need = (2 * per + (size_t)B * T * V) * sizeof(double) + 2 * per_ex * sizeof(int32_t);
        void* raw = nullptr;                    // alpha | beta mantissas (fp64), the softmax (fp64), then the two exponent arrays: per-stream scratch owned by the library
        if (int e = stream_scratch(SCRATCH_CTC, s, need, &raw)) return e;
        double* ws = reinterpret_cast<double*>(raw);
        a.y_ws = ws + 2 * per;
        if (grad) {
            a.alpha_ws = ws;
            a.beta_ws = ws + per;
            a.alpha_ex = reinterpret_cast<int32_t*>(a.y_ws + (size_t)B * T * V);
            a.beta_ex = a.alpha_ex + per_ex;
        }
    }
    const size_t lds = (size_t)(4 * (U + 3)) * sizeof(double) + (size_t)(((2 * (U + 3) + 3) & ~3) + ((U + 1 + 3) & ~3)) * sizeof(int) + 16;
    W2V2_REQUIRE(U + 1 <= CTC_THREADS, "ctc: %d labels per row; this build holds one state pair per thread, up to %d", U, CTC_THREADS - 1);
    W2V2_REQUIRE(lds <= 60 * 1024, "ctc: U=%d needs %zu B of LDS", U, lds);          // (under the 64 KiB a kernel gets without opting in)
    ProfScope ps(prof, FAM_CTC, 30.0 * B * (double)T * (2 * U + 1), 4.0 * B * (double)T * V * (grad ? 2 : 1), s);
    W2V2_LAUNCH(ctc_softmax_kernel, dim3((unsigned)(((int64_t)B * T + 3) / 4)), dim3(256), 0, s, a);
    W2V2_LAUNCH(ctc_kernel, dim3(B, grad ? 2 : 1), dim3(CTC_THREADS), lds, s, a);
    if (grad) W2V2_LAUNCH(ctc_grad_kernel, dim3(T, B), dim3(64), (size_t)V * (sizeof(unsigned long long) + sizeof(double)), s, a);
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
