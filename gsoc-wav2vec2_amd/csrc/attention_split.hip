// Fused self-attention with fp32-level results on the bf16 matrix cores (precision mode W2V2_PRECISION_BF16X3,
// inference forward, head size 64).
//
// The algorithm, the "transposed tile" layout (S^T = K Q^T so that a lane owns one query column; O^T = V^T P^T taking
// P straight from the S^T accumulator registers) and the LDS images are those of attention_bf16.hip.  What differs is
// the arithmetic: nothing is rounded to bf16.  Every fp32 operand -- q d^-0.5, k, v and the probabilities -- is split
// exactly into three bf16 terms, x = x0 + x1 + x2, and each contraction keeps the six bf16 x bf16 products of order <= 2
// (gemm_split.hip explains the error bound: below one fp32 ulp of each product), accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16, smallest terms first.  Scores, max, exp (the compensated form of attention.hip), sums and
// the output are fp32.  96 MFMAs per 32 queries x 64 keys = 3072 matrix cycles against 16 x 2 x 64 = 2048 for the fp32
// MFMA kernel's 16-pass instructions -- the win is that the fp32 pipe is 16x slower per flop, the split 6x more flops.
//
// Block = 8 waves x 32 queries; K and V^T live in LDS as three bf16 planes each (6 x 8 KiB per stage, two stages = 96 KiB,
// one block per CU, two waves per SIMD).  The next tile is loaded to registers under the current tile's work, split
// and stored after it.
//
// FMT = PF_F16X2 (precision mode f16x2): the same kernel with every operand as TWO fp16 terms and three products per contraction
// (a1 b0 + a0 b1 + a0 b0) -- half the MFMAs, two planes per LDS image, a cheaper split.  Operands are scaled by powers of two into
// fp16's comfortable range (q d^-0.5, k, v by F16X2_ACT_SCALE like every activation of the mode; the probabilities, which lie in
// [0, 1], by 2^10) and the accumulators rescaled exactly; a value beyond the range saturates and sets the mode's sticky flag.
#include "common.h"

namespace w2v2 {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int DH = 64;      // head size
constexpr int KT = 64;      // keys per tile
constexpr int NW = 8;       // waves per block, 32 queries each
constexpr int NT = NW * 64;
constexpr int ROWB = 128;   // bytes per LDS row: 64 bf16
constexpr int PLANE = KT * ROWB;        // one 16-bit plane of K (64 keys x 64 d) or of V^T (64 d x 64 keys): 8 KiB
constexpr int stage_bytes(int fmt) { return 2 * plane_count(fmt) * PLANE; }      // K planes, then V^T planes
constexpr float P_SCALE = 1024.0f;      // f16x2: probabilities x 2^10 before their split
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

struct AttnSplitArgs {
    const float* qkv;           // (B, T, 3H): q | k | v
    const int32_t* frame_len;   // (B) or null
    float* ctx;                 // (B, T, H), or null when only the planes are wanted
    int B, T, H, heads;
    float scale;
    PlaneOut planes;            // optional planes of ctx for the out-projection GEMM (gemm_split_sw.hip)
    int* range_flag;            // f16x2: sticky saturation flag (may be null)
};


// two fp32 -> one dword per plane (lo = first value), exact three-term split
struct Split3 { unsigned p0, p1, p2; };
__device__ __forceinline__ Split3 split2(float a, float b) {
    Split3 r;
    r.p0 = pack_bf16_rne(a, b);
    a -= __uint_as_float(r.p0 << 16);
    b -= __uint_as_float(r.p0 & 0xffff0000u);
    r.p1 = pack_bf16_rne(a, b);
    a -= __uint_as_float(r.p1 << 16);
    b -= __uint_as_float(r.p1 & 0xffff0000u);
    r.p2 = pack_bf16_rne(a, b);
    return r;
}
#define W2V2_SPLIT_INTO(x, y, v0, v1, v2) \
    do { const Split3 t_ = split2((x), (y)); (v0) = t_.p0; (v1) = t_.p1; (v2) = t_.p2; } while (0)

// two fp32 (already scaled) -> one dword per plane of the format: bf16x3 exact three-term split | f16x2 two fp16 terms, saturating
template <int FMT>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&pl)[3], bool& ovf) {
    if constexpr (FMT == PF_F16X2) {
        ovf |= !(fabsf(a) <= F16X2_MAX) | !(fabsf(b) <= F16X2_MAX);
        a = __builtin_amdgcn_fmed3f(a, -F16X2_MAX, F16X2_MAX);
        b = __builtin_amdgcn_fmed3f(b, -F16X2_MAX, F16X2_MAX);
        const h2_t h = {(_Float16)a, (_Float16)b};
        pl[0] = __builtin_bit_cast(unsigned, h);
        pl[1] = pack_f16_rne(a - (float)h[0], b - (float)h[1]);
        pl[2] = 0u;
    } else {
        const Split3 t = split2(a, b);
        pl[0] = t.p0; pl[1] = t.p1; pl[2] = t.p2;
    }
}
template <int FMT>
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (FMT == PF_F16X2)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// exp(x), x <= 0, with the rounding of x * log2(e) compensated (see attention.hip::exp_compensated)
__device__ __forceinline__ float exp_comp(float x) {
    constexpr float L2E_HI = 1.44269504088896340736f, L2E_LO = 1.925963033500822e-08f, LN2 = 0.69314718055994530942f;
    x = fmaxf(x, -1.0e30f);
    const float y = x * L2E_HI;
    const float e = fmaf(x, L2E_LO, fmaf(x, L2E_HI, -y));
    return __builtin_amdgcn_exp2f(y) * fmaf(e, LN2, 1.0f);
}

// swizzles of attention_bf16.hip: 16-byte slots of the K image (rows = keys), 8-byte slots of the V^T image (rows = d)
__device__ __forceinline__ int swz_k(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 1); }
__device__ __forceinline__ int swz_v(int d) { return (((d >> 1) & 7) << 1) ^ ((d >> 4) & 1); }

// the (first operand plane, second operand plane) pairs of order <= 2, smallest products first: six for three planes, three for two
template <int FMT> struct Terms;
template <> struct Terms<PF_BF16X3> { static constexpr int N = 6; static constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0}; };
template <> struct Terms<PF_F16X2> { static constexpr int N = 3; static constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0}; };

template <int FMT>
__global__ __launch_bounds__(NT, 2) void attention_split_kernel(AttnSplitArgs a) {
    constexpr int NP = plane_count(FMT), STAGE = stage_bytes(FMT), NTERM = Terms<FMT>::N;
    constexpr float OPS = FMT == PF_F16X2 ? F16X2_ACT_SCALE : 1.0f;       // scale of q d^-0.5, k, v before their split
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_as[];
    bool ovf = false;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const int64_t ld = 3 * (int64_t)a.H;
    const float* __restrict__ base = a.qkv + (int64_t)b * a.T * ld + head * DH;
    const int flen = a.frame_len ? a.frame_len[b] : a.T;

    // ---- Q fragments (B operand of S^T), three planes: lane = (query li, half lh), d = 16 st + 8 lh .. + 7, pre-scaled ----
    u32x4 qf[NP][4];
    {
        const int qr = min(q0 + li, a.T - 1);
        const float* qp = base + (int64_t)qr * ld + 8 * lh;
        const float qs = a.scale * OPS;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(qp + 16 * st);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(qp + 16 * st + 4);
            const float e[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned pl[3];
                split_pair<FMT>(e[2 * j] * qs, e[2 * j + 1] * qs, pl, ovf);
#pragma unroll
                for (int p = 0; p < NP; ++p) qf[p][st][j] = pl[p];
            }
        }
    }

    // ---- staging: K as 2 float4 per thread (16 lanes = one 256-byte row), V as a 4-key x 2-d patch per thread ----
    f32x4 rk[2];
    f32x2 rv[4];
    const int v_dp = tid & 31, v_c = tid >> 5;          // d pair, 4-key chunk (= 8-byte slot of the V^T row)
    auto load_tile = [&](int tile) {
        const int k0 = tile * KT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * NT, r = idx >> 4, sl = idx & 15;
            const int key = min(k0 + r, a.T - 1);       // clamp: tail rows are masked out below
            rk[i] = *reinterpret_cast<const f32x4*>(base + (int64_t)key * ld + a.H + sl * 4);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int key = min(k0 + 4 * v_c + kk, a.T - 1);
            rv[kk] = *reinterpret_cast<const f32x2*>(base + (int64_t)key * ld + 2 * a.H + 2 * v_dp);
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* S = smem_as + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * NT, r = idx >> 4, sl = idx & 15;
            unsigned lo[3], hi[3];
            split_pair<FMT>(rk[i][0] * OPS, rk[i][1] * OPS, lo, ovf);
            split_pair<FMT>(rk[i][2] * OPS, rk[i][3] * OPS, hi, ovf);
            unsigned char* dst = S + r * ROWB + (((sl >> 1) ^ swz_k(r)) << 4) + (sl & 1) * 8;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(dst + p * PLANE) = u32x2{lo[p], hi[p]};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {                    // register transpose: column j of the patch = 4 consecutive keys
            const int d = 2 * v_dp + j;
            unsigned lo[3], hi[3];
            split_pair<FMT>(rv[0][j] * OPS, rv[1][j] * OPS, lo, ovf);
            split_pair<FMT>(rv[2][j] * OPS, rv[3][j] * OPS, hi, ovf);
            unsigned char* dst = S + NP * PLANE + d * ROWB + ((v_c ^ swz_v(d)) << 3);
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(dst + p * PLANE) = u32x2{lo[p], hi[p]};
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (a.T + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * KT, buf = tile & 1;
        load_tile(tile + 1 < ntiles ? tile + 1 : tile);     // unconditional (the last one re-reads)
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* Ks = smem_as + buf * STAGE;
        const unsigned char* Vs = Ks + NP * PLANE;

        // ---- S^T = K Q^T for two 32-key sub-tiles: 2 x 4 x 6 MFMAs, the two accumulators interleaved ----
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            u32x4 kf[2][NP];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int row = kt * 32 + li;
                const unsigned char* kp = Ks + row * ROWB + (((2 * st + lh) ^ swz_k(row)) << 4);
#pragma unroll
                for (int p = 0; p < NP; ++p) kf[kt][p] = *reinterpret_cast<const u32x4*>(kp + p * PLANE);
            }
#pragma unroll
            for (int t = 0; t < NTERM; ++t)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    s[kt] = mfma16<FMT>(kf[kt][Terms<FMT>::PA[t]], qf[Terms<FMT>::PB[t]][st], s[kt]);
        }
        if constexpr (FMT == PF_F16X2) {       // undo the operand scales (exact)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt][r] *= 1.0f / (OPS * OPS);
        }
        // ---- mask + online softmax (lane owns query li; keys (r&3) + 8 (r>>2) + 4 lh) ----
        if (k0 + KT > min(flen, a.T)) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = s[kt][r];
                    v = key >= flen ? v - 10000.0f : v;       // (1 - mask) * -10000, encoder.py:256-257
                    v = key >= a.T ? -INFINITY : v;           // tile padding: not a key at all
                    s[kt][r] = v;
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);          // exp(-inf) = 0 on the first tile
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = exp_comp(s[kt][r] - m_new);
                s[kt][r] = p;
                rs += p;
            }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

        // ---- O^T += V^T P^T: 2 x 2 x 2 x 6 MFMAs.  B = the accumulator registers 8h .. 8h+7 of the lane, split into three
        // planes and packed pairwise (keys 16 h + {0..3, 8..11} + 4 lh of the sub-tile); A = two 8-byte reads per plane of the
        // V^T image at the same keys ----
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 pb[NP];
                constexpr float PS = FMT == PF_F16X2 ? P_SCALE : 1.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned pl[3];
                    bool never = false;                   // (probabilities x 2^10 <= 1024: cannot saturate)
                    split_pair<FMT>(s[kt][8 * h + 2 * j] * PS, s[kt][8 * h + 2 * j + 1] * PS, pl, never);
#pragma unroll
                    for (int p = 0; p < NP; ++p) pb[p][j] = pl[p];
                }
                u32x4 vf[2][NP];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int d = dt * 32 + li, sw = swz_v(d);
                    const unsigned char* row = Vs + d * ROWB;
                    const int oa = ((8 * kt + 4 * h + lh) ^ sw) << 3, ob = ((8 * kt + 4 * h + 2 + lh) ^ sw) << 3;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const u32x2 va = *reinterpret_cast<const u32x2*>(row + p * PLANE + oa);
                        const u32x2 vb = *reinterpret_cast<const u32x2*>(row + p * PLANE + ob);
                        vf[dt][p] = u32x4{va[0], va[1], vb[0], vb[1]};
                    }
                }
#pragma unroll
                for (int t = 0; t < NTERM; ++t)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        o[dt] = mfma16<FMT>(vf[dt][Terms<FMT>::PA[t]], pb[Terms<FMT>::PB[t]], o[dt]);
            }
        __builtin_amdgcn_sched_barrier(0);
        store_tile(buf ^ 1);        // the other stage was last read one iteration ago (barrier below closed it)
        __syncthreads();
    }

    // ---- normalise and store: O^T rows are d = 32 dt + (r&3) + 8 (r>>2) + 4 lh, column = query ----
    const int q = q0 + li;
    report_overflow(a.range_flag, ovf);
    ovf = false;
    if (q < a.T) {
        const float inv = (1.0f / l_run) * (FMT == PF_F16X2 ? 1.0f / (P_SCALE * OPS) : 1.0f);      // (f16x2: O carries the scales of P and V)
        const int64_t off = ((int64_t)b * a.T + q) * a.H + head * DH + 4 * lh;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {o[d][4 * g] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv};
                if (a.ctx) *reinterpret_cast<f32x4*>(a.ctx + off + 32 * d + 8 * g) = v;
                if (a.planes.p) store_planes4(a.planes.p + off + 32 * d + 8 * g, a.planes.plane, a.planes.fmt, v, ovf);
            }
    }
    report_overflow(a.planes.range_flag, ovf);
}

}  // namespace

bool attention_split_supported(int head_dim) { return head_dim == DH; }

int launch_attention_split(const float* qkv, const int32_t* frame_len, float* ctx, int B, int T, int H, int heads,
                           hipStream_t s, const PlaneOut* planes, int fmt, int* range_flag) {
    const PlaneOut pl = planes ? *planes : PlaneOut{};
    W2V2_REQUIRE(qkv && (ctx || pl.p) && B > 0 && T > 0 && heads > 0, "attention_split: bad argument");
    W2V2_REQUIRE(!pl.p || (pl.plane % 4 == 0 && (reinterpret_cast<uintptr_t>(pl.p) & 7) == 0), "attention_split: unaligned planes");
    W2V2_REQUIRE(H / heads == DH && H % heads == 0, "attention_split: head size %d unsupported (64)", H / heads);
    W2V2_REQUIRE((H % 4) == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0 && (reinterpret_cast<uintptr_t>(ctx) & 15) == 0,
                 "attention_split: unaligned buffers");
    W2V2_REQUIRE(fmt == PF_BF16X3 || fmt == PF_F16X2, "attention_split: unknown plane format %d", fmt);
    AttnSplitArgs a{qkv, frame_len, ctx, B, T, H, heads, 1.0f / sqrtf((float)DH), pl, range_flag};
    static std::atomic<bool> attr_set{false};   // (idempotent call; atomic so concurrent host threads agree on the flag)
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_split_kernel<PF_BF16X3>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * stage_bytes(PF_BF16X3)));
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_split_kernel<PF_F16X2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * stage_bytes(PF_F16X2)));
        attr_set = true;
    }
    dim3 grid((T + NW * 32 - 1) / (NW * 32), heads, B), block(NT);
    if (fmt == PF_F16X2)
        W2V2_LAUNCH(attention_split_kernel<PF_F16X2>, grid, block, 2 * stage_bytes(PF_F16X2), s, a);
    else
        W2V2_LAUNCH(attention_split_kernel<PF_BF16X3>, grid, block, 2 * stage_bytes(PF_BF16X3), s, a);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace w2v2
