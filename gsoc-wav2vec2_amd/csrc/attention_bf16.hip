// Fused self-attention on the bf16 matrix pipe (precision mode W2V2_PRECISION_BF16, head size 64).
//
// Same algorithm and the same "transposed tile" layout trick as attention.hip -- S^T = K Q^T so that a lane
// owns one query column, online softmax in fp32 registers, O^T = V^T P^T taking P straight from the S^T
// accumulator registers -- with v_mfma_f32_32x32x16_bf16 doing the two contractions: q (pre-scaled,
// encoder.py:28), k, v and the (dropped-out) probabilities are rounded to bf16 (nearest-even), products are
// exact, accumulation, max / exp / sum and the output stay fp32.
//
// What changes with the 16-deep MFMA:
//   * S^T: A = K rows (ds_read_b128 of 8 consecutive d), B = Q fragments held in registers (16 VGPRs).
//   * P as the B operand of the second contraction: accumulator registers r = 8h .. 8h+7 of the lane (query,
//     half lh) are the keys 16h + {0..3, 8..11} + 4 lh of the 32-key sub-tile, so packing them pairwise
//     (v_cvt_pk_bf16_f32) IS the 8-element B fragment, provided the A operand (V^T) supplies the same keys in
//     the same positions: two 8-byte reads of a TRANSPOSED V image  Vt[d][key]  at keys 16h + 4lh and
//     16h + 8 + 4lh.  V is transposed on the way into LDS (a thread owns an 8-key x 2-d patch), like the B
//     operand of gemm_bf16.hip.
//   * K and V come from the fp32 packed qkv buffer through registers (round + store), one tile prefetched
//     under the current tile's work; both LDS images are XOR-swizzled so every access is bank-conflict free.
// The matrix work is 16x cheaper than in fp32 (16 MFMAs = 512 cycles per 32 queries x 64 keys), so the
// kernel is bound by the softmax VALU work (and, in training, the dropout hash); it runs small 4-wave blocks,
// several per CU, to cover that with other waves' MFMAs.
#include "common.h"
#include "train.h"

namespace w2v2 {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int DH = 64;    // head size (768 / 12 = 1024 / 16 = 64 for every published checkpoint)
constexpr int KT = 64;    // keys per tile
constexpr int NW = 4;     // waves per block, 32 queries each
constexpr int ROWB = 128; // bytes per LDS row: 64 bf16
constexpr float LOG2E = 1.44269504088896340736f;

struct Attn16Args {
    const float* qkv;           // (B, T, 3H): q | k | v
    const int32_t* frame_len;   // (B) or null
    float* ctx;                 // (B, T, H)
    uint16_t* ctx16;            // optional bf16 shadow of ctx (the out-projection GEMM's A operand)
    int B, T, H, heads;
    float scale;
};

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// 16-byte-slot swizzle of the K image (rows = keys): conflict-free ds_read_b128 over 32 consecutive rows
__device__ __forceinline__ int swz_k(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 1); }
// 8-byte-slot swizzle of the Vt image (rows = d): conflict-free ds_read_b64 over 32 consecutive rows and
// conflict-free transposing stores (16 lanes = 16 row pairs)
__device__ __forceinline__ int swz_v(int d) { return (((d >> 1) & 7) << 1) ^ ((d >> 4) & 1); }

template <bool TRAIN>
__global__ __launch_bounds__(NW * 64, 2) void attention_bf16_kernel(Attn16Args a, AttnTrain tr) {
    constexpr int STAGE = 2 * KT * ROWB;     // K image (64 keys x 128 B) + Vt image (64 d x 128 B)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_a16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const int64_t ld = 3 * (int64_t)a.H;
    const float* __restrict__ base = a.qkv + (int64_t)b * a.T * ld + head * DH;
    const int flen = a.frame_len ? a.frame_len[b] : a.T;

    // ---- Q fragments (B operand of S^T): lane = (query li, half lh), d = 16 st + 8 lh .. + 7, pre-scaled ----
    u32x4 qf[4];
    {
        const int qr = min(q0 + li, a.T - 1);
        const float* qp = base + (int64_t)qr * ld + 8 * lh;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(qp + 16 * st);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(qp + 16 * st + 4);
            qf[st][0] = pack_bf16(v0[0] * a.scale, v0[1] * a.scale);
            qf[st][1] = pack_bf16(v0[2] * a.scale, v0[3] * a.scale);
            qf[st][2] = pack_bf16(v1[0] * a.scale, v1[1] * a.scale);
            qf[st][3] = pack_bf16(v1[2] * a.scale, v1[3] * a.scale);
        }
    }

    // ---- staging: K as 4 float4 per thread (16 lanes = one 256-byte row), V as an 8-key x 2-d patch ----
    f32x4 rk[4];
    f32x2 rv[8];
    const int v_dp = tid & 31, v_c = tid >> 5;          // d pair, 8-key chunk
    auto load_tile = [&](int tile) {
        const int k0 = tile * KT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, r = idx >> 4, sl = idx & 15;
            const int key = min(k0 + r, a.T - 1);       // clamp: tail rows are masked out below
            rk[i] = *reinterpret_cast<const f32x4*>(base + (int64_t)key * ld + a.H + sl * 4);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int key = min(k0 + 8 * v_c + kk, a.T - 1);
            rv[kk] = *reinterpret_cast<const f32x2*>(base + (int64_t)key * ld + 2 * a.H + 2 * v_dp);
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* S = smem_a16 + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, r = idx >> 4, sl = idx & 15;
            u32x2 p;
            p[0] = pack_bf16(rk[i][0], rk[i][1]);
            p[1] = pack_bf16(rk[i][2], rk[i][3]);
            *reinterpret_cast<u32x2*>(S + r * ROWB + (((sl >> 1) ^ swz_k(r)) << 4) + (sl & 1) * 8) = p;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {                    // register transpose: column j of the patch = 8 consecutive keys
            const int d = 2 * v_dp + j, sw = swz_v(d);
            u32x2 lo, hi;
            lo[0] = pack_bf16(rv[0][j], rv[1][j]);
            lo[1] = pack_bf16(rv[2][j], rv[3][j]);
            hi[0] = pack_bf16(rv[4][j], rv[5][j]);
            hi[1] = pack_bf16(rv[6][j], rv[7][j]);
            unsigned char* row = S + KT * ROWB + d * ROWB;
            *reinterpret_cast<u32x2*>(row + (((2 * v_c) ^ sw) << 3)) = lo;
            *reinterpret_cast<u32x2*>(row + (((2 * v_c + 1) ^ sw) << 3)) = hi;
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    // dropout hash inputs hoisted out of the tile loop: element index = ((b h + head) T + q) T + key, modulo 2^32
    const uint32_t drop_key = TRAIN ? dropout_key(tr.seed, tr.stream) : 0u, drop_thr = TRAIN ? dropout_threshold(tr.p) : 0u;
    const uint32_t drop_row = (uint32_t)((((uint64_t)b * a.heads + head) * a.T + (uint64_t)min(q0 + li, a.T - 1)) * a.T);

    const int ntiles = (a.T + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * KT, buf = tile & 1;
        load_tile(tile + 1 < ntiles ? tile + 1 : tile);     // unconditional (the last one re-reads): keeps the staging set in registers
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* Ks = smem_a16 + buf * STAGE;
        const unsigned char* Vs = Ks + KT * ROWB;

        // ---- S^T = K Q^T for two 32-key sub-tiles: 8 MFMAs ----
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
            const int row = kt * 32 + li;
            const unsigned char* kp = Ks + row * ROWB;
            const int sw = swz_k(row);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(kp + (((2 * st + lh) ^ sw) << 4));
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(kf), as_bf16x8(qf[st]), s[kt], 0, 0, 0);
            }
        }
        // ---- mask + online softmax (lane owns query li; keys (r&3) + 8 (r>>2) + 4 lh) ----
        // The kernel is VALU-bound (the 16 MFMAs of a tile are 512 cycles), so the per-score work is kept minimal:
        // masking only on tiles that touch the valid-length / T boundary (wave-uniform branch), exp as one FMA + v_exp_f32.
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
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);   // exp2(-inf) = 0 on the first tile
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f((s[kt][r] - m_new) * LOG2E);   // subtract first: exact for nearby values even at |m| = 1e4 (masked rows)
                s[kt][r] = p;
                rs += p;
            }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        if (TRAIN && tr.p > 0.f) {
            // attention-probability dropout (encoder.py:42-44): the row sum above uses the un-dropped p; the
            // 1 / (1 - p) factor is applied once, with the final normalisation
            const uint32_t cbase = drop_row + (uint32_t)k0;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t col = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    s[kt][r] = dropout_keep32(drop_key, cbase + col, drop_thr) ? s[kt][r] : 0.f;
                }
        }
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

        // ---- O^T += Vt P^T: 8 MFMAs; B = the packed accumulator registers, A = two 8-byte reads of Vt ----
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 pb;
#pragma unroll
                for (int j = 0; j < 4; ++j) pb[j] = pack_bf16(s[kt][8 * h + 2 * j], s[kt][8 * h + 2 * j + 1]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int d = dt * 32 + li, sw = swz_v(d);
                    const unsigned char* row = Vs + d * ROWB;
                    const u32x2 va = *reinterpret_cast<const u32x2*>(row + (((8 * kt + 4 * h + lh) ^ sw) << 3));
                    const u32x2 vb = *reinterpret_cast<const u32x2*>(row + (((8 * kt + 4 * h + 2 + lh) ^ sw) << 3));
                    const u32x4 vf = {va[0], va[1], vb[0], vb[1]};
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vf), as_bf16x8(pb), o[dt], 0, 0, 0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        store_tile(buf ^ 1);        // the other stage was last read one iteration ago (barrier below closed it)
        __syncthreads();
    }

    // ---- normalise and store: O^T rows are d = 32 dt + (r&3) + 8 (r>>2) + 4 lh, column = query ----
    const int q = q0 + li;
    if (TRAIN && q < a.T && lh == 0) tr.lse[((int64_t)b * a.heads + head) * a.T + q] = m_run + logf(l_run);
    if (q < a.T) {
        const float inv = ((TRAIN && tr.p > 0.f) ? 1.0f / (1.0f - tr.p) : 1.0f) / l_run;
        float* op = a.ctx + ((int64_t)b * a.T + q) * a.H + head * DH + 4 * lh;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(op + 32 * d + 8 * g) =
                    f32x4{o[d][4 * g] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv};
        if (a.ctx16) {      // bf16 shadow for the out-projection GEMM
            uint16_t* hp = a.ctx16 + ((int64_t)b * a.T + q) * a.H + head * DH + 4 * lh;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<u32x2*>(hp + 32 * d + 8 * g) =
                        u32x2{pack_bf16(o[d][4 * g] * inv, o[d][4 * g + 1] * inv), pack_bf16(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv)};
        }
    }
}


// ======================================================================================
// Backward (see attention.hip for the math).  Same ownership as the fp32 kernels -- a wave owns 32 queries
// (dQ) or 32 keys (dK, dV) whose [col][d] fragments sit in registers as B operands, the other side streams
// through LDS -- but every streamed tile is kept in LDS TWICE: row-major [row][d] for the contractions over d
// (S, dP) and transposed [d][row] for the contractions over the streamed rows (dQ, dK, dV), because the
// 16-deep MFMA wants 8 consecutive k per lane.  dS and P(dropped) go from the accumulator registers to the B
// operand by pairwise packing, exactly as P does in the forward.
// Operands rounded to bf16: q d^-0.5, k, v, dO, dS, P keep/(1-p).  fp32: scores, exp, D, dS arithmetic, sums.
// ======================================================================================
struct Attn16BwdArgs {
    const float* qkv;
    const int32_t* frame_len;
    const float* d_o;       // (B, T, H)
    const float* dvec;      // (B, heads, T)
    float* dqkv;            // (B, T, 3H)
    uint16_t* dqkv16;       // optional bf16 shadow of dqkv (the A operand of the q|k|v data-gradient GEMM)
    int B, T, H, heads;
    float scale;
};

// 64 rows x 64 d fp32 -> registers, row mapping (16 lanes = one 256-byte row) and patch mapping (8 rows x 2 d)
__device__ __forceinline__ void load_rows(f32x4 (&n)[4], const float* src, int64_t ld, int r0, int T, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256, r = idx >> 4, sl = idx & 15;
        n[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)min(r0 + r, T - 1) * ld + sl * 4);
    }
}
__device__ __forceinline__ void load_patch(f32x2 (&p)[8], const float* src, int64_t ld, int r0, int T, int tid) {
    const int dp = tid & 31, c = tid >> 5;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
        p[kk] = *reinterpret_cast<const f32x2*>(src + (int64_t)min(r0 + 8 * c + kk, T - 1) * ld + 2 * dp);
}
// -> [row][d] image (16-byte slots swizzled by swz_k)
__device__ __forceinline__ void store_rows(const f32x4 (&n)[4], unsigned char* img, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256, r = idx >> 4, sl = idx & 15;
        u32x2 p;
        p[0] = pack_bf16(n[i][0], n[i][1]);
        p[1] = pack_bf16(n[i][2], n[i][3]);
        *reinterpret_cast<u32x2*>(img + r * ROWB + (((sl >> 1) ^ swz_k(r)) << 4) + (sl & 1) * 8) = p;
    }
}
// -> [d][row] image (8-byte slots swizzled by swz_v)
__device__ __forceinline__ void store_patch(const f32x2 (&p)[8], unsigned char* img, int tid) {
    const int dp = tid & 31, c = tid >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int d = 2 * dp + j, sw = swz_v(d);
        u32x2 lo, hi;
        lo[0] = pack_bf16(p[0][j], p[1][j]);
        lo[1] = pack_bf16(p[2][j], p[3][j]);
        hi[0] = pack_bf16(p[4][j], p[5][j]);
        hi[1] = pack_bf16(p[6][j], p[7][j]);
        unsigned char* row = img + d * ROWB;
        *reinterpret_cast<u32x2*>(row + (((2 * c) ^ sw) << 3)) = lo;
        *reinterpret_cast<u32x2*>(row + (((2 * c + 1) ^ sw) << 3)) = hi;
    }
}
// the same patch -> the [row][d] image as well (8 dword stores: d pair 2 dp of rows 8 c + kk), so a tile that is
// needed in both orientations is loaded from global once and held in 16 registers
__device__ __forceinline__ void store_patch_rows(const f32x2 (&p)[8], unsigned char* img, int tid) {
    const int dp = tid & 31, c = tid >> 5;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int r = 8 * c + kk;
        *reinterpret_cast<unsigned*>(img + r * ROWB + (((dp >> 2) ^ swz_k(r)) << 4) + (dp & 3) * 4) = pack_bf16(p[kk][0], p[kk][1]);
    }
}
// A fragment of a contraction over d: 8 consecutive d of `row`, k-step st, lane half lh
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* img, int row, int st, int lh) {
    return as_bf16x8(*reinterpret_cast<const u32x4*>(img + row * ROWB + (((2 * st + lh) ^ swz_k(row)) << 4)));
}
// A fragment of a contraction over the streamed rows: the rows that accumulator registers 8h .. 8h+7 of
// sub-tile t hold, for output row d
__device__ __forceinline__ bf16x8 frag_cols(const unsigned char* img, int d, int t, int h, int lh) {
    const int sw = swz_v(d);
    const unsigned char* row = img + d * ROWB;
    const u32x2 va = *reinterpret_cast<const u32x2*>(row + (((8 * t + 4 * h + lh) ^ sw) << 3));
    const u32x2 vb = *reinterpret_cast<const u32x2*>(row + (((8 * t + 4 * h + 2 + lh) ^ sw) << 3));
    return as_bf16x8(u32x4{va[0], va[1], vb[0], vb[1]});
}
__device__ __forceinline__ bf16x8 pack_acc(const f32x16& v, int h) {
    u32x4 pb;
#pragma unroll
    for (int j = 0; j < 4; ++j) pb[j] = pack_bf16(v[8 * h + 2 * j], v[8 * h + 2 * j + 1]);
    return as_bf16x8(pb);
}
// [col][d] register fragments (B operand of the contractions over d), optionally scaled before rounding
__device__ __forceinline__ void load_col_frags(u32x4 (&f)[4], const float* p, float scale) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + 16 * st);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + 16 * st + 4);
        f[st][0] = pack_bf16(v0[0] * scale, v0[1] * scale);
        f[st][1] = pack_bf16(v0[2] * scale, v0[3] * scale);
        f[st][2] = pack_bf16(v1[0] * scale, v1[1] * scale);
        f[st][3] = pack_bf16(v1[2] * scale, v1[3] * scale);
    }
}

// ---- dQ: block = 4 waves x 32 queries; streams 64-key tiles: K [key][d], V [key][d], Kt [d][key] ----
__global__ __launch_bounds__(256, 2) void attention_bf16_bwd_dq_kernel(Attn16BwdArgs a, AttnTrain tr) {
    constexpr int IMG = KT * ROWB, STAGE = 3 * IMG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_a16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const int64_t ld = 3 * (int64_t)a.H;
    const float* __restrict__ base = a.qkv + (int64_t)b * a.T * ld + head * DH;
    const int flen = a.frame_len ? a.frame_len[b] : a.T;
    const int qr = min(q0 + li, a.T - 1);
    const bool qok = q0 + li < a.T;

    u32x4 qf[4], dof[4];
    load_col_frags(qf, base + (int64_t)qr * ld + 8 * lh, a.scale);
    load_col_frags(dof, a.d_o + ((int64_t)b * a.T + qr) * a.H + head * DH + 8 * lh, 1.0f);
    const int64_t sidx = ((int64_t)b * a.heads + head) * a.T + qr;
    const float lse = tr.lse[sidx], dv = a.dvec[sidx];
    const float inv = tr.p > 0.f ? 1.0f / (1.0f - tr.p) : 1.0f;
    const uint64_t rowbase = (uint64_t)sidx * a.T;
    const uint32_t drop_key = dropout_key(tr.seed, tr.stream), drop_thr = dropout_threshold(tr.p);

    f32x16 dq[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

    f32x4 rv[4];
    f32x2 rkt[8];
    auto load_tile = [&](int tile) {
        load_rows(rv, base + 2 * a.H, ld, tile * KT, a.T, tid);
        load_patch(rkt, base + a.H, ld, tile * KT, a.T, tid);
    };
    auto store_tile = [&](int buf) {
        unsigned char* S = smem_a16 + buf * STAGE;
        store_patch_rows(rkt, S, tid);
        store_rows(rv, S + IMG, tid);
        store_patch(rkt, S + 2 * IMG, tid);
    };
    const int ntiles = (a.T + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * KT, buf = tile & 1;
        load_tile(tile + 1 < ntiles ? tile + 1 : tile);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* Ks = smem_a16 + buf * STAGE;
        const unsigned char* Vs = Ks + IMG;
        const unsigned char* Kt = Ks + 2 * IMG;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
            const int row = kt * 32 + li;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ks, row, st, lh), as_bf16x8(qf[st]), s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Vs, row, st, lh), as_bf16x8(dof[st]), dp, 0, 0, 0);
            }
            // dS^T = P^T * (dP^T * keep/(1-p) - D);  P = exp(S - lse) = exp2((S - lse) log2e)
            if (k0 + KT > min(flen, a.T)) {         // boundary tiles only (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float sv = s[r];
                    sv = key >= flen ? sv - 10000.0f : sv;
                    s[r] = key < a.T ? sv : -INFINITY;       // exp2(-inf) = 0: a padding key contributes nothing
                }
            }
            const uint32_t cbase = (uint32_t)rowbase + (uint32_t)(k0 + kt * 32 + 4 * lh);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f((s[r] - lse) * LOG2E);
                float g = dp[r];
                if (tr.p > 0.f) g = dropout_keep32(drop_key, cbase + (uint32_t)((r & 3) + 8 * (r >> 2)), drop_thr) ? g : 0.f;
                s[r] = pv * fmaf(g, inv, -dv);
            }
            // dQ^T[d][q] += sum_key Kt[d][key] dS^T[key][q]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8 pb = pack_acc(s, h);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(Kt, dt * 32 + li, kt, h, lh), pb, dq[dt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        store_tile(buf ^ 1);
        __syncthreads();
    }
    if (qok) {
        const int64_t o0 = ((int64_t)b * a.T + q0 + li) * ld + head * DH + 4 * lh;
        float* op = a.dqkv + o0;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = f32x4{dq[d][4 * g] * a.scale, dq[d][4 * g + 1] * a.scale, dq[d][4 * g + 2] * a.scale, dq[d][4 * g + 3] * a.scale};
                *reinterpret_cast<f32x4*>(op + 32 * d + 8 * g) = v;
                if (a.dqkv16)
                    *reinterpret_cast<u32x2*>(a.dqkv16 + o0 + 32 * d + 8 * g) = u32x2{pack_bf16_rne(v[0], v[1]), pack_bf16_rne(v[2], v[3])};
            }
    }
}

// ---- dK, dV: block = 4 waves x 32 keys; streams 64-query tiles: Q, dO as [q][d] and as [d][q], + lse, D ----
__global__ __launch_bounds__(256, 2) void attention_bf16_bwd_dkv_kernel(Attn16BwdArgs a, AttnTrain tr) {
    constexpr int IMG = KT * ROWB, STAGE = 4 * IMG + 2 * KT * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_a16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int c0 = (blockIdx.x * NW + wave) * 32;          // this wave's 32 keys
    const int64_t ld = 3 * (int64_t)a.H;
    const float* __restrict__ base = a.qkv + (int64_t)b * a.T * ld + head * DH;
    const float* __restrict__ dobase = a.d_o + (int64_t)b * a.T * a.H + head * DH;
    const int flen = a.frame_len ? a.frame_len[b] : a.T;
    const int key = c0 + li;
    const int kr = min(key, a.T - 1);
    const bool kok = key < a.T;
    const float kmask = key >= flen ? -10000.0f : 0.0f;
    const uint32_t drop_key = dropout_key(tr.seed, tr.stream), drop_thr = dropout_threshold(tr.p);

    u32x4 kf[4], vf[4];
    load_col_frags(kf, base + (int64_t)kr * ld + a.H + 8 * lh, a.scale);       // S = scale q.k (scale is a power of two)
    load_col_frags(vf, base + (int64_t)kr * ld + 2 * a.H + 8 * lh, 1.0f);
    const float inv = tr.p > 0.f ? 1.0f / (1.0f - tr.p) : 1.0f;
    const int64_t bh = (int64_t)b * a.heads + head;
    const uint32_t drop_col = (uint32_t)((uint64_t)bh * a.T * a.T) + (uint32_t)kr;

    f32x16 dk[2], dvv[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[d][r] = dvv[d][r] = 0.f;

    f32x2 rqt[8], rot[8];
    float rl = 0.f;
    auto load_tile = [&](int tile) {
        load_patch(rqt, base, ld, tile * KT, a.T, tid);
        load_patch(rot, dobase, a.H, tile * KT, a.T, tid);
        if (tid < 2 * KT) {
            const int qq = min(tile * KT + (tid & (KT - 1)), a.T - 1);
            rl = tid < KT ? tr.lse[bh * a.T + qq] : a.dvec[bh * a.T + qq];
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* S = smem_a16 + buf * STAGE;
        store_patch_rows(rqt, S, tid);
        store_patch_rows(rot, S + IMG, tid);
        store_patch(rqt, S + 2 * IMG, tid);
        store_patch(rot, S + 3 * IMG, tid);
        if (tid < 2 * KT) reinterpret_cast<float*>(S + 4 * IMG)[tid] = rl;
    };
    const int ntiles = (a.T + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int t0 = tile * KT, buf = tile & 1;
        load_tile(tile + 1 < ntiles ? tile + 1 : tile);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* Qs = smem_a16 + buf * STAGE;
        const unsigned char* Os = Qs + IMG;
        const unsigned char* Qt = Qs + 2 * IMG;
        const unsigned char* Ot = Qs + 3 * IMG;
        const float* Ls = reinterpret_cast<const float*>(Qs + 4 * IMG);      // [0, KT): lse, [KT, 2KT): D
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
            const int row = qt * 32 + li;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qs, row, st, lh), as_bf16x8(kf[st]), s, 0, 0, 0);     // S[q][key]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Os, row, st, lh), as_bf16x8(vf[st]), dp, 0, 0, 0);   // dP[q][key]
            }
            // lane owns key column `key`; register r is query row qt*32 + (r&3) + 8 (r>>2) + 4 lh
            // (columns of keys >= T are clamped duplicates whose results are never stored, so only the QUERY bound
            // needs masking, and only on the last tile)
            const bool qtail = t0 + KT > a.T;
            uint32_t didx = drop_col + (uint32_t)(t0 + qt * 32 + 4 * lh) * (uint32_t)a.T;   // ((bh T + q) T + key) mod 2^32
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float pv = __builtin_amdgcn_exp2f((s[r] + kmask - Ls[ql]) * LOG2E);
                if (qtail) pv = t0 + ql < a.T ? pv : 0.f;
                float g = dp[r], pd = pv;
                if (tr.p > 0.f) {
                    const bool keep = dropout_keep32(drop_key, didx + (uint32_t)((r & 3) + 8 * (r >> 2)) * (uint32_t)a.T, drop_thr);
                    g = keep ? g : 0.f;
                    pd = keep ? pv * inv : 0.f;
                }
                dp[r] = pd;                                    // Pd[q][key]
                s[r] = pv * fmaf(g, inv, -Ls[KT + ql]);        // dS[q][key]
            }
            // dV^T[d][key] += sum_q dOt[d][q] Pd[q][key];   dK^T[d][key] += sum_q Qt[d][q] dS[q][key]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8 pbp = pack_acc(dp, h), pbs = pack_acc(s, h);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dvv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(Ot, dt * 32 + li, qt, h, lh), pbp, dvv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(Qt, dt * 32 + li, qt, h, lh), pbs, dk[dt], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        store_tile(buf ^ 1);
        __syncthreads();
    }
    if (kok) {
        const int64_t k0 = ((int64_t)b * a.T + key) * ld + a.H + head * DH + 4 * lh;
        float* kp = a.dqkv + k0;
        float* vp = kp + a.H;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // kf was pre-scaled for S, so dS is the gradient of the SCALED score: dK = scale * dS^T Q
                const f32x4 kv = f32x4{dk[d][4 * g] * a.scale, dk[d][4 * g + 1] * a.scale, dk[d][4 * g + 2] * a.scale, dk[d][4 * g + 3] * a.scale};
                const f32x4 vv = f32x4{dvv[d][4 * g], dvv[d][4 * g + 1], dvv[d][4 * g + 2], dvv[d][4 * g + 3]};
                *reinterpret_cast<f32x4*>(kp + 32 * d + 8 * g) = kv;
                *reinterpret_cast<f32x4*>(vp + 32 * d + 8 * g) = vv;
                if (a.dqkv16) {
                    *reinterpret_cast<u32x2*>(a.dqkv16 + k0 + 32 * d + 8 * g) = u32x2{pack_bf16_rne(kv[0], kv[1]), pack_bf16_rne(kv[2], kv[3])};
                    *reinterpret_cast<u32x2*>(a.dqkv16 + k0 + a.H + 32 * d + 8 * g) = u32x2{pack_bf16_rne(vv[0], vv[1]), pack_bf16_rne(vv[2], vv[3])};
                }
            }
    }
}

}  // namespace

bool attention_bf16_supported(int head_size) { return head_size == DH; }

// tr == nullptr: inference.  Otherwise the training forward (dropout on P, lse saved).
int launch_attention_fwd_bf16(const float* qkv, const int32_t* frame_len, float* ctx, uint16_t* ctx16, int B, int T, int H,
                              int heads, const AttnTrain* tr, hipStream_t s) {
    W2V2_REQUIRE(H / heads == DH, "attention_bf16: head size %d unsupported (64)", H / heads);
    Attn16Args a{qkv, frame_len, ctx, ctx16, B, T, H, heads, 1.0f / sqrtf((float)DH)};
    const size_t lds = 2 * 2 * KT * ROWB;
    dim3 grid((T + NW * 32 - 1) / (NW * 32), heads, B), block(NW * 64);
    if (tr)
        hipLaunchKernelGGL(attention_bf16_kernel<true>, grid, block, lds, s, a, *tr);
    else
        hipLaunchKernelGGL(attention_bf16_kernel<false>, grid, block, lds, s, a, AttnTrain{0.f, 0, 0, nullptr});
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

// D must already be in `dvec` (attn_dvec_kernel, fp32, launched by the caller)
int launch_attention_bwd_bf16(const float* qkv, const int32_t* frame_len, const float* dctx, const float* dvec,
                              float* dqkv, uint16_t* dqkv16, int B, int T, int H, int heads, const AttnTrain& tr, hipStream_t s) {
    W2V2_REQUIRE(H / heads == DH, "attention_bwd_bf16: head size %d unsupported (64)", H / heads);
    W2V2_REQUIRE((reinterpret_cast<uintptr_t>(dqkv16) & 7) == 0, "attention_bwd_bf16: unaligned shadow");
    Attn16BwdArgs a{qkv, frame_len, dctx, dvec, dqkv, dqkv16, B, T, H, heads, 1.0f / sqrtf((float)DH)};
    const size_t lds_q = 2 * 3 * KT * ROWB, lds_kv = 2 * (4 * KT * ROWB + 2 * KT * 4);
    static std::atomic<bool> attr_set{false};   // (idempotent call; atomic so concurrent host threads agree on the flag)
    if (!attr_set) {
        W2V2_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bf16_bwd_dkv_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv));
        attr_set = true;
    }
    dim3 grid((T + 127) / 128, heads, B), block(256);
    hipLaunchKernelGGL(attention_bf16_bwd_dq_kernel, grid, block, lds_q, s, a, tr);
    hipLaunchKernelGGL(attention_bf16_bwd_dkv_kernel, grid, block, lds_kv, s, a, tr);
    W2V2_HIP_CHECK(hipGetLastError());
    return W2V2_OK;
}

}  // namespace w2v2
