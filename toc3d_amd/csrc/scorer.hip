// Motion-aware query-guided token scorer and the remaining small ops of the ToC3D backbone (gfx950).
//
// Reference: MotionAwareQueryGuidedTokenSelector.get_motion_aware_queries (backbones/toc3d_utils.py:334-360),
// NaiveQueryGuidedTokenSelector.query_based_score (:232-252), ScoreBasedTokenSelector.score/sample (:114-158),
// MLN / transform_reference_points (utils/misc.py:154-200), pos2posemb3d / pos2posemb1d /
// nerf_positional_encoding (utils/positional_encoding.py:14-81), get_abs_pos (backbones/eva_utils.py:229-258),
// CPFPN 3x3 conv input (necks/cp_fpn.py:124-133).
//
// Everything here is f32 (f64 for the time embedding): the scores decide a discrete top-k, so no reduced
// precision.  The per-token scorer is HBM-bound (one read of x); the query side is 64 x 256 numbers and is
// latency-bound, so it is fused into one launch per stage with weights pre-transposed for coalesced reads.
// No fast-math intrinsics: sin/cos arguments reach 5e10 and need full range reduction.
#include "capi.h"
#include "common.h"

namespace {

constexpr int QD = 256;        // query dim (toc3d_utils.py:198)
constexpr int MD = 180;        // NeRF-encoded ego-motion dim (15 x 6 x 2), toc3d_utils.py:327
constexpr int PE3 = 384;       // pos2posemb3d width (3 x 128)

// packed scorer weights (floats); every matrix is stored transposed [in][out] with out = 256
struct MW {
    static constexpr int qe0_w = 0;                       // [384][256]
    static constexpr int qe0_b = qe0_w + PE3 * QD;
    static constexpr int qe2_w = qe0_b + QD;              // [256][256]
    static constexpr int qe2_b = qe2_w + QD * QD;
    static constexpr int pe_red_w = qe2_b + QD;           // [180][256]
    static constexpr int pe_red_b = pe_red_w + MD * QD;
    static constexpr int pe_gam_w = pe_red_b + QD;
    static constexpr int pe_gam_b = pe_gam_w + QD * QD;
    static constexpr int pe_bet_w = pe_gam_b + QD;
    static constexpr int pe_bet_b = pe_bet_w + QD * QD;
    static constexpr int q_red_w = pe_bet_b + QD;
    static constexpr int q_red_b = q_red_w + MD * QD;
    static constexpr int q_gam_w = q_red_b + QD;
    static constexpr int q_gam_b = q_gam_w + QD * QD;
    static constexpr int q_bet_w = q_gam_b + QD;
    static constexpr int q_bet_b = q_bet_w + QD * QD;
    static constexpr int te_w = q_bet_b + QD;
    static constexpr int te_b = te_w + QD * QD;
    static constexpr int te_ln_w = te_b + QD;
    static constexpr int te_ln_b = te_ln_w + QD;
    static constexpr int pc_range = te_ln_b + QD;         // [6] (+2 pad)
    static constexpr int dimt3 = pc_range + 8;            // [128] temperature ** (2*floor(i/2)/128), as torch computes it
    static constexpr int dimt1 = dimt3 + 128;             // [256] temperature ** (2*floor(i/2)/256)
    static constexpr int total = dimt1 + 256;
};

__global__ void transpose_copy_kernel(const float* __restrict__ src, int out_dim, int in_dim, float* __restrict__ dst) {
    const int n = out_dim * in_dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int o = i % out_dim, k = i / out_dim;       // dst[k][o] = src[o][k]
        dst[i] = src[(int64_t)o * in_dim + k];
    }
}

TOC3D_DEV float block_sum256(float v, float* s_red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// ---- motion-aware queries: one 1024-thread workgroup per QB queries.  Thread (t, part): t = output feature (256),
// part = the input indices i = part (mod 4) it accumulates, for all QB queries at once: each weight is fetched once
// per workgroup and used QB times (the matvecs were L2-bandwidth bound at one query per workgroup: 2.3 MB of weights
// each).  Inputs sit in LDS as [i][QB] so one 32-byte read feeds the QB FMAs of a weight; weights are fetched sixteen
// rows ahead of the FMAs.  Per query the arithmetic (FMA order, the fixed-order 4-way combine, the LN reductions) is
// the same as with one query per workgroup, so results do not depend on QB or on the grouping.
constexpr int QB = 8;

// R weight rows (i, i+4, ...) fetched together, then their FMAs in row order
template <int R>
TOC3D_DEV void matvec_rows(const float* __restrict__ Wt, const float (*in)[QB], int i, int t, float (&acc)[QB]) {
    float wv[R];
#pragma unroll
    for (int u = 0; u < R; ++u) wv[u] = Wt[(int64_t)(i + 4 * u) * QD + t];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(&in[i + 4 * u][0]), hi = *reinterpret_cast<const f32x4*>(&in[i + 4 * u][4]);
        acc[0] = fmaf(lo.x, wv[u], acc[0]); acc[1] = fmaf(lo.y, wv[u], acc[1]); acc[2] = fmaf(lo.z, wv[u], acc[2]); acc[3] = fmaf(lo.w, wv[u], acc[3]);
        acc[4] = fmaf(hi.x, wv[u], acc[4]); acc[5] = fmaf(hi.y, wv[u], acc[5]); acc[6] = fmaf(hi.z, wv[u], acc[6]); acc[7] = fmaf(hi.w, wv[u], acc[7]);
    }
}

TOC3D_DEV void matvec256(const float* __restrict__ Wt, const float* __restrict__ b, const float (*in)[QB], int n_in, int t, int part,
                         float (*s_mv)[4][QD], float (&res)[QB]) {
    static_assert(QB == 8, "matvec_rows reads the eight queries of a row as two float4");
    float acc[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) acc[q] = 0.f;
    int i = part;
    for (; i + 4 * 15 < n_in; i += 4 * 16) matvec_rows<16>(Wt, in, i, t, acc);
    if (i + 4 * 7 < n_in) { matvec_rows<8>(Wt, in, i, t, acc); i += 4 * 8; }
    if (i + 4 * 3 < n_in) { matvec_rows<4>(Wt, in, i, t, acc); i += 4 * 4; }
    for (; i < n_in; i += 4) matvec_rows<1>(Wt, in, i, t, acc);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < QB; ++q) s_mv[q][part][t] = acc[q];
    __syncthreads();
    const float bt = b[t];
#pragma unroll
    for (int q = 0; q < QB; ++q) res[q] = ((s_mv[q][0][t] + s_mv[q][1][t]) + (s_mv[q][2][t] + s_mv[q][3][t])) + bt;
}

// per query: sum over the 256 features (each held identically by the 4 `part` copies; only part 0 contributes)
TOC3D_DEV void feat_sum(const float (&v)[QB], int part, float (*s_red)[4], float (&out)[QB]) {
    float w[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) w[q] = wave_sum(part == 0 ? v[q] : 0.f);
    __syncthreads();
    if (part == 0 && (threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < QB; ++q) s_red[q][threadIdx.x >> 6] = w[q];                     // part 0 = waves 0..3
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < QB; ++q) out[q] = (s_red[q][0] + s_red[q][1]) + (s_red[q][2] + s_red[q][3]);
}

TOC3D_DEV void ln256_noaffine(const float (&v)[QB], float eps, int part, float (*s_red)[4], float (&out)[QB]) {
    float m[QB], d[QB], dd[QB], var[QB];
    feat_sum(v, part, s_red, m);
#pragma unroll
    for (int q = 0; q < QB; ++q) { d[q] = v[q] - m[q] * (1.0f / QD); dd[q] = d[q] * d[q]; }
    feat_sum(dd, part, s_red, var);
#pragma unroll
    for (int q = 0; q < QB; ++q) out[q] = d[q] * (1.0f / sqrtf(var[q] * (1.0f / QD) + eps));
}

__global__ __launch_bounds__(1024) void motion_queries_kernel(const float* __restrict__ w_all, int64_t w_stride, const float* __restrict__ queries,
                                                              const float* __restrict__ ref, const float* __restrict__ vel,
                                                              const void* __restrict__ ts, int ts_f64, const float* __restrict__ pose,
                                                              const float* __restrict__ pose_inv, int Q, int BQ, int groups,
                                                              float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float s_emb[PE3][QB];
    __shared__ __attribute__((aligned(16))) float s_h[QD][QB];
    __shared__ __attribute__((aligned(16))) float s_e[MD + 12][QB];
    __shared__ float s_mv[QB][4][QD];
    __shared__ float s_red[QB][4];
    __shared__ float s_pts[QB][4];
    __shared__ float s_ego[QB][16];
    const int stage = blockIdx.x / groups, bq0 = (blockIdx.x % groups) * QB;
    const int t = threadIdx.x & 255, part = threadIdx.x >> 8;
    const float* w = w_all + (int64_t)stage * w_stride;
    out += (int64_t)stage * BQ * QD;
    const float kTorchLnEps = 1e-5f;
    const float two_pi = 6.283185307179586f;              // float(2 * math.pi)
    // a short last group repeats its last query in the unused lanes (computed, never written)
    auto query_of = [&](int q) { return min(bq0 + q, BQ - 1); };

    // 1. reference points -> current ego frame -> normalised by pc_range (misc.py:191-200, toc3d_utils.py:346-348)
    if (threadIdx.x < 4 * QB && (threadIdx.x & 3) < 3) {
        const int q = threadIdx.x >> 2, c = threadIdx.x & 3, bq = query_of(q);
        const float* m = pose_inv + (int64_t)(bq / Q) * 16 + c * 4;
        const float* p = ref + (int64_t)bq * 3;
        const float v = ((m[0] * p[0] + m[1] * p[1]) + m[2] * p[2]) + m[3];
        const float* pc = w + MW::pc_range;
        s_pts[q][c] = (v - pc[c]) / (pc[3 + c] - pc[c]);
    }
    // 3a. ego-motion vector [vel(2), t(1), pose[:3,:](12)] as f32 (toc3d_utils.py:351)
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 16 * QB && ((threadIdx.x - 64) & 15) < 15) {
        const int q = (threadIdx.x - 64) >> 4, j = (threadIdx.x - 64) & 15, bq = query_of(q);
        float v;
        if (j < 2) v = vel[(int64_t)bq * 2 + j];
        else if (j == 2) v = ts_f64 ? (float)reinterpret_cast<const double*>(ts)[bq] : reinterpret_cast<const float*>(ts)[bq];
        else v = pose[(int64_t)bq * 16 + (j - 3)];
        s_ego[q][j] = v;
    }
    __syncthreads();
    // 2. pos2posemb3d, concatenated (y, x, z) (positional_encoding.py:14-26)
    for (int e = threadIdx.x; e < PE3 * QB; e += 1024) {
        const int i = e / QB, q = e % QB;
        const int blk = i >> 7, f = i & 127;
        const int coord = blk == 0 ? 1 : (blk == 1 ? 0 : 2);
        const float a = (s_pts[q][coord] * two_pi) / w[MW::dimt3 + f];
        s_emb[i][q] = (f & 1) ? cosf(a) : sinf(a);
    }
    // 3b. NeRF encoding, frequency-major: [sin(2^k e), cos(2^k e)] for k = 0..5 (positional_encoding.py:73-75)
    for (int e = threadIdx.x; e < MD * QB; e += 1024) {
        const int i = e / QB, q = e % QB;
        const int k = i / 30, r = i % 30;
        const float a = s_ego[q][r % 15] * (float)(1 << k);
        s_e[i][q] = r < 15 ? sinf(a) : cosf(a);
    }
    __syncthreads();
    float pos[QB], tmp[QB], gam[QB], bet[QB], nrm[QB];
    // query_embedding: Linear(384,256) - ReLU - Linear(256,256) (toc3d_utils.py:322-326,349)
    matvec256(w + MW::qe0_w, w + MW::qe0_b, s_emb, PE3, t, part, s_mv, tmp);
    __syncthreads();
    if (part == 0) {
#pragma unroll
        for (int q = 0; q < QB; ++q) s_h[t][q] = fmaxf(tmp[q], 0.f);
    }
    __syncthreads();
    matvec256(w + MW::qe2_w, w + MW::qe2_b, s_h, QD, t, part, s_mv, pos);
    // MLN over pos (misc.py:181-188)
    {
        matvec256(w + MW::pe_red_w, w + MW::pe_red_b, s_e, MD, t, part, s_mv, tmp);
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int q = 0; q < QB; ++q) s_h[t][q] = fmaxf(tmp[q], 0.f);
        }
        __syncthreads();
        matvec256(w + MW::pe_gam_w, w + MW::pe_gam_b, s_h, QD, t, part, s_mv, gam);
        matvec256(w + MW::pe_bet_w, w + MW::pe_bet_b, s_h, QD, t, part, s_mv, bet);
        ln256_noaffine(pos, kTorchLnEps, part, s_red, nrm);
#pragma unroll
        for (int q = 0; q < QB; ++q) pos[q] = gam[q] * nrm[q] + bet[q];
    }
    // time embedding: pos2posemb1d in the timestamp's dtype (f64 when the head promoted it), then .float()
    {
        const float dt = w[MW::dimt1 + t];
        __syncthreads();
        {
#pragma unroll 1
            for (int qq = 0; qq < QB / 4; ++qq) {                // two queries' embeddings per thread
                const int q = part * (QB / 4) + qq, bq = query_of(q);
                float e;
                if (ts_f64) {
                    const double a = (reinterpret_cast<const double*>(ts)[bq] * 6.283185307179586) / (double)dt;
                    e = (float)((t & 1) ? cos(a) : sin(a));
                } else {
                    const float a = (reinterpret_cast<const float*>(ts)[bq] * two_pi) / dt;
                    e = (t & 1) ? cosf(a) : sinf(a);
                }
                s_emb[t][q] = e;
            }
        }
        __syncthreads();
        matvec256(w + MW::te_w, w + MW::te_b, s_emb, QD, t, part, s_mv, tmp);
        ln256_noaffine(tmp, kTorchLnEps, part, s_red, nrm);
        const float lw = w[MW::te_ln_w + t], lb = w[MW::te_ln_b + t];
#pragma unroll
        for (int q = 0; q < QB; ++q) pos[q] += nrm[q] * lw + lb;
    }
    // MLN over the memory queries, then add pos (toc3d_utils.py:356-358)
    {
        matvec256(w + MW::q_red_w, w + MW::q_red_b, s_e, MD, t, part, s_mv, tmp);
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int q = 0; q < QB; ++q) s_h[t][q] = fmaxf(tmp[q], 0.f);
        }
        __syncthreads();
        matvec256(w + MW::q_gam_w, w + MW::q_gam_b, s_h, QD, t, part, s_mv, gam);
        matvec256(w + MW::q_bet_w, w + MW::q_bet_b, s_h, QD, t, part, s_mv, bet);
#pragma unroll
        for (int q = 0; q < QB; ++q) tmp[q] = queries[(int64_t)query_of(q) * QD + t];
        ln256_noaffine(tmp, kTorchLnEps, part, s_red, nrm);
        if (part == 0) {
#pragma unroll
            for (int q = 0; q < QB; ++q)
                if (bq0 + q < BQ) out[(int64_t)(bq0 + q) * QD + t] = (gam[q] * nrm[q] + bet[q]) + pos[q];
        }
    }
}

// Wc[b][i][j] = scale * sum_c W_in[c][i] * u[c][j],  u[c][j] = sum_q mq[b][q][c] * W_agg[j][q]
__global__ __launch_bounds__(256) void collapse_kernel(const float* __restrict__ mq, const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                       const float* __restrict__ w_agg, const float* __restrict__ b_agg, int Q, int C, float scale,
                                                       float* __restrict__ wc, float* __restrict__ bc) {
    __shared__ float s_u[QD][2];
    __shared__ float s_red[4];
    const int b = blockIdx.y, t = threadIdx.x;
    float u0 = 0.f, u1 = 0.f;
    int q = 0;
    for (; q + 16 <= Q; q += 16) {               // sixteen rows in flight; the FMA order is the plain q order
        float m[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) m[u] = mq[((int64_t)b * Q + q + u) * QD + t];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            u0 = fmaf(m[u], w_agg[q + u], u0);
            u1 = fmaf(m[u], w_agg[Q + q + u], u1);
        }
    }
    for (; q < Q; ++q) {
        const float m = mq[((int64_t)b * Q + q) * QD + t];
        u0 = fmaf(m, w_agg[q], u0);
        u1 = fmaf(m, w_agg[Q + q], u1);
    }
    s_u[t][0] = u0;
    s_u[t][1] = u1;
    __syncthreads();
    const int i = blockIdx.x * 256 + t;
    if (i < C) {
        float a0 = 0.f, a1 = 0.f;
        for (int c = 0; c < QD; c += 16) {
            float wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = w_in[(int64_t)(c + u) * C + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a0 = fmaf(wv[u], s_u[c + u][0], a0);
                a1 = fmaf(wv[u], s_u[c + u][1], a1);
            }
        }
        wc[((int64_t)b * C + i) * 2 + 0] = a0 * scale;
        wc[((int64_t)b * C + i) * 2 + 1] = a1 * scale;
    }
    if (blockIdx.x == 0) {
        const float bi = b_in[t];
        const float s0 = block_sum256(bi * u0, s_red);
        const float s1 = block_sum256(bi * u1, s_red);
        if (t == 0) {
            bc[b * 2 + 0] = s0 * scale + b_agg[0];
            bc[b * 2 + 1] = s1 * scale + b_agg[1];
        }
    }
}

// log_softmax over 2 logits + Gumbel soft mask (toc3d_utils.py:111,147)
TOC3D_DEV void score_tail(float l0, float l1, const float* gumbel, int64_t row, float* pred, float* score, float* mask_out) {
    const float mx = fmaxf(l0, l1);
    const float lse = logf(expf(l0 - mx) + expf(l1 - mx));
    const float p0 = (l0 - mx) - lse, p1 = (l1 - mx) - lse;
    pred[row * 2] = p0;
    pred[row * 2 + 1] = p1;
    score[row] = p0;
    const float a0 = p0 + (gumbel ? gumbel[row * 2] : 0.f), a1 = p1 + (gumbel ? gumbel[row * 2 + 1] : 0.f);
    const float am = fmaxf(a0, a1);
    const float e0 = expf(a0 - am), e1 = expf(a1 - am);
    mask_out[row] = e0 / (e0 + e1);
}

__global__ __launch_bounds__(256) void score_tokens_kernel(const float* __restrict__ x, int C, const float* __restrict__ mask,
                                                           const float* __restrict__ wc, const float* __restrict__ bc,
                                                           const float* __restrict__ gumbel, int64_t M, int T, int vpf,
                                                           float* __restrict__ pred, float* __restrict__ score, float* __restrict__ mask_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= M) return;
    const int b = (int)(row / T) / vpf;
    const float mk = mask ? mask[row] : 1.f;
    const float* w = wc + (int64_t)b * C * 2;
    float d0 = 0.f, d1 = 0.f;
    for (int vi = lane; vi < (C >> 2); vi += 64) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * C + 4 * vi);
        const f32x4 wa = *reinterpret_cast<const f32x4*>(w + 8 * vi), wb = *reinterpret_cast<const f32x4*>(w + 8 * vi + 4);
        const float x0 = xv[0] * mk, x1 = xv[1] * mk, x2 = xv[2] * mk, x3 = xv[3] * mk;
        d0 += (x0 * wa[0] + x1 * wa[2]) + (x2 * wb[0] + x3 * wb[2]);
        d1 += (x0 * wa[1] + x1 * wa[3]) + (x2 * wb[1] + x3 * wb[3]);
    }
    d0 = wave_sum(d0);
    d1 = wave_sum(d1);
    if (lane == 0) score_tail(d0 + bc[b * 2], d1 + bc[b * 2 + 1], gumbel, row, pred, score, mask_out);
}

template <typename T>
__global__ __launch_bounds__(256) void score_head_kernel(const T* __restrict__ f, int64_t ld, int kdim, const float* __restrict__ w,
                                                         const float* __restrict__ b, const float* __restrict__ gumbel, int64_t M,
                                                         float* __restrict__ pred, float* __restrict__ score, float* __restrict__ mask_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= M) return;
    float d0 = 0.f, d1 = 0.f;
    for (int c = lane * 8; c < kdim; c += 512) {
        float v[8];
        load8(f + row * ld + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            d0 = fmaf(v[e], w[c + e], d0);
            d1 = fmaf(v[e], w[kdim + c + e], d1);
        }
    }
    d0 = wave_sum(d0);
    d1 = wave_sum(d1);
    if (lane == 0) score_tail(d0 + b[0], d1 + b[1], gumbel, row, pred, score, mask_out);
}

// columns [C/2, C) of each view <- mean over the view's tokens (toc3d_utils.py:125-126)
template <typename T>
__global__ __launch_bounds__(256) void global_mean_half_kernel(T* __restrict__ t, int64_t ld, int Tn, int C) {
    __shared__ float s_p[4][64];
    const int v = blockIdx.y, cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = C / 2 + blockIdx.x * 64 + cl;
    const bool ok = c < C;
    float s = 0.f;
    if (ok)
        for (int r = rl; r < Tn; r += 4) s += from_act(t[((int64_t)v * Tn + r) * ld + c]);
    s_p[rl][cl] = s;
    __syncthreads();
    const float mean = ((s_p[0][cl] + s_p[1][cl]) + (s_p[2][cl] + s_p[3][cl])) / (float)Tn;
    if (ok)
        for (int r = rl; r < Tn; r += 4) t[((int64_t)v * Tn + r) * ld + c] = to_act<T>(mean);
}

// torch F.interpolate(mode='bicubic', align_corners=False): cubic convolution, A = -0.75, border-clamped taps
TOC3D_DEV void cubic_coeffs(float t, float (&c)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x3 = 2.f - t, x2 = 1.f - t;
    c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    c[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__global__ __launch_bounds__(256) void abs_pos_bicubic_kernel(const float* __restrict__ pos, int S, int C, float* __restrict__ out, int h, int w) {
    const int y = blockIdx.x / w, x = blockIdx.x % w;
    const float sy = (float)S / (float)h, sx = (float)S / (float)w;
    const float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float cy[4], cx[4];
    cubic_coeffs(fy - (float)iy, cy);
    cubic_coeffs(fx - (float)ix, cx);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = min(max(iy - 1 + i, 0), S - 1);
            float r = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = min(max(ix - 1 + j, 0), S - 1);
                r += pos[((int64_t)yy * S + xx) * C + c] * cx[j];
            }
            acc += r * cy[i];
        }
        out[(int64_t)blockIdx.x * C + c] = acc;
    }
}

// 3x3, pad 1 im2col over NHWC f32: row (v,y,x), col (ky,kx,c)
template <typename T>
__global__ __launch_bounds__(256) void im2col_3x3_kernel(const float* __restrict__ x, T* __restrict__ out, int64_t ldo, int V, int h, int w, int C) {
    const int64_t m = blockIdx.x;
    const int xx = (int)(m % w), yy = (int)((m / w) % h), v = (int)(m / ((int64_t)w * h));
    for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) {
        const int tap = i / C, c = i % C;
        const int y2 = yy + tap / 3 - 1, x2 = xx + tap % 3 - 1;
        const float val = (y2 >= 0 && y2 < h && x2 >= 0 && x2 < w) ? x[(((int64_t)v * h + y2) * w + x2) * C + c] : 0.f;
        out[m * ldo + i] = to_act<T>(val);
    }
}

// ---- Gumbel noise of the soft masks, drawn on the device INSIDE the recorded frame (toc3d_utils.py:145-147: F.gumbel_softmax samples
// -log(E), E ~ Exp(1), i.e. -log(-log(U))).  Counter-based: Philox4x32-10 keyed by the model's seed, counter = (frame counter, element / 4), so a
// replayed launch plan draws fresh noise every frame although its launch arguments never change: the frame counter lives in device memory
// (state[0]) and is advanced by the last workgroup of the launch (ticket in state[1], self-resetting).
TOC3D_DEV void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// 32 random bits -> one Gumbel(0, 1) sample -log(-log(U)).  U = (23 random bits + 0.5) * 2^-23: the sum has 24 significant bits, so it is exact in f32
// and U lies in [2^-24, 1 - 2^-24] -- strictly inside (0, 1), both logs finite for EVERY input word.  (The first form took 24 bits + 0.5: 25
// significant bits, and 0xFFFFFF + 0.5 rounded to 2^24, i.e. U = 1 and a sample of +inf once in 2^24 draws -> NaN soft mask.)
TOC3D_DEV float gumbel_from_bits(unsigned bits) {
    const float u = ((float)(bits >> 9) + 0.5f) * 1.1920928955078125e-07f;
    return -logf(-logf(u));
}

__global__ __launch_bounds__(256) void gumbel_from_bits_kernel(const unsigned* __restrict__ bits, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = gumbel_from_bits(bits[i]);
}

__global__ __launch_bounds__(256) void gumbel_noise_kernel(float* __restrict__ out, int64_t n, unsigned long long seed, unsigned long long* __restrict__ state) {
    const unsigned long long frame = state[0];
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one Philox block = 4 values
    if (q * 4 < n) {
        unsigned c[4] = {(unsigned)q, (unsigned)(q >> 32), (unsigned)frame, (unsigned)(frame >> 32)};
        philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
        float gv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] = gumbel_from_bits(c[e]);
        if (q * 4 + 4 <= n) *reinterpret_cast<f32x4*>(out + q * 4) = f32x4{gv[0], gv[1], gv[2], gv[3]};
        else for (int e = 0; q * 4 + e < n; ++e) out[q * 4 + e] = gv[e];
    }
    // every workgroup has read state[0] before it takes its ticket; the last one to arrive advances the frame counter
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned long long t = atomicAdd(&state[1], 1ull);
        if (t == (unsigned long long)gridDim.x - 1) { state[1] = 0; state[0] = frame + 1; __threadfence(); }
    }
}

}  // namespace

extern "C" {

int64_t toc3d_motion_weights_floats(void) { return MW::total; }

int toc3d_pack_motion_weights(const float* qe0_w, const float* qe0_b, const float* qe2_w, const float* qe2_b,
                              const float* pe_red_w, const float* pe_red_b, const float* pe_gam_w, const float* pe_gam_b,
                              const float* pe_bet_w, const float* pe_bet_b, const float* q_red_w, const float* q_red_b,
                              const float* q_gam_w, const float* q_gam_b, const float* q_bet_w, const float* q_bet_b,
                              const float* te_w, const float* te_b, const float* te_ln_w, const float* te_ln_b,
                              const float* pc_range, const float* dimt3, const float* dimt1, float* out, toc3d_stream_t stream) {
    TOC3D_REQUIRE(qe0_w && qe0_b && qe2_w && qe2_b && pe_red_w && pe_red_b && pe_gam_w && pe_gam_b && pe_bet_w && pe_bet_b && q_red_w &&
                      q_red_b && q_gam_w && q_gam_b && q_bet_w && q_bet_b && te_w && te_b && te_ln_w && te_ln_b && pc_range && dimt3 && dimt1 && out,
                  "toc3d_pack_motion_weights: null buffer");
    hipStream_t s = as_stream(stream);
    struct M { const float* src; int out_dim, in_dim, off; };
    const M mats[] = {{qe0_w, QD, PE3, MW::qe0_w}, {qe2_w, QD, QD, MW::qe2_w}, {pe_red_w, QD, MD, MW::pe_red_w}, {pe_gam_w, QD, QD, MW::pe_gam_w},
                      {pe_bet_w, QD, QD, MW::pe_bet_w}, {q_red_w, QD, MD, MW::q_red_w}, {q_gam_w, QD, QD, MW::q_gam_w}, {q_bet_w, QD, QD, MW::q_bet_w},
                      {te_w, QD, QD, MW::te_w}};
    for (const M& m : mats)
        toc3d_launch(transpose_copy_kernel, dim3(64), dim3(256), 0, s, m.src, m.out_dim, m.in_dim, out + m.off);
    struct Vv { const float* src; int n, off; };
    const Vv vecs[] = {{qe0_b, QD, MW::qe0_b}, {qe2_b, QD, MW::qe2_b}, {pe_red_b, QD, MW::pe_red_b}, {pe_gam_b, QD, MW::pe_gam_b},
                       {pe_bet_b, QD, MW::pe_bet_b}, {q_red_b, QD, MW::q_red_b}, {q_gam_b, QD, MW::q_gam_b}, {q_bet_b, QD, MW::q_bet_b},
                       {te_b, QD, MW::te_b}, {te_ln_w, QD, MW::te_ln_w}, {te_ln_b, QD, MW::te_ln_b}, {pc_range, 6, MW::pc_range},
                       {dimt3, 128, MW::dimt3}, {dimt1, 256, MW::dimt1}};
    for (const Vv& v : vecs)
        toc3d_launch(transpose_copy_kernel, dim3(1), dim3(256), 0, s, v.src, v.n, 1, out + v.off);
    TOC3D_LAUNCH_CHECK("toc3d_pack_motion_weights");
    return TOC3D_OK;
}

int toc3d_motion_queries(const float* w, int64_t n_stages, int64_t w_stride, const float* queries, const float* ref_points, const float* vel,
                         const void* timestamp, int timestamp_is_f64, const float* ego_pose, const float* ego_pose_inv, int64_t B, int64_t Q,
                         float* out, toc3d_stream_t stream) {
    TOC3D_REQUIRE(w && queries && ref_points && vel && timestamp && ego_pose && ego_pose_inv && out, "toc3d_motion_queries: null buffer");
    TOC3D_REQUIRE(n_stages >= 1 && (n_stages == 1 || w_stride >= MW::total), "toc3d_motion_queries: bad n_stages / w_stride");
    if (B <= 0 || Q <= 0) return TOC3D_OK;
    const int64_t groups = (B * Q + QB - 1) / QB;
    toc3d_launch(motion_queries_kernel, dim3((unsigned)(n_stages * groups)), dim3(1024), 0, as_stream(stream), w, w_stride, queries, ref_points, vel,
                       timestamp, timestamp_is_f64, ego_pose, ego_pose_inv, (int)Q, (int)(B * Q), (int)groups, out);
    TOC3D_LAUNCH_CHECK("toc3d_motion_queries");
    return TOC3D_OK;
}

int toc3d_collapse_query_scorer(const float* mq, const float* w_in, const float* b_in, const float* w_agg, const float* b_agg,
                                int64_t B, int64_t Q, int64_t C, float scale, float* wc, float* bc, toc3d_stream_t stream) {
    TOC3D_REQUIRE(mq && w_in && b_in && w_agg && b_agg && wc && bc, "toc3d_collapse_query_scorer: null buffer");
    if (B <= 0) return TOC3D_OK;
    dim3 grid((unsigned)((C + 255) / 256), (unsigned)B);
    toc3d_launch(collapse_kernel, grid, dim3(256), 0, as_stream(stream), mq, w_in, b_in, w_agg, b_agg, (int)Q, (int)C, scale, wc, bc);
    TOC3D_LAUNCH_CHECK("toc3d_collapse_query_scorer");
    return TOC3D_OK;
}

int toc3d_score_tokens(const float* x, int64_t C, const float* mask, const float* wc, const float* bc, const float* gumbel,
                       int64_t V, int64_t T, int64_t views_per_frame, float* pred, float* score, float* mask_out,
                       toc3d_stream_t stream) {
    TOC3D_REQUIRE(x && wc && bc && pred && score && mask_out, "toc3d_score_tokens: null buffer");
    TOC3D_REQUIRE(C % 4 == 0 && views_per_frame > 0 && V % views_per_frame == 0, "toc3d_score_tokens: bad dims");
    const int64_t M = V * T;
    if (M <= 0) return TOC3D_OK;
    toc3d_launch(score_tokens_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, as_stream(stream), x, (int)C, mask, wc, bc, gumbel, M, (int)T,
                       (int)views_per_frame, pred, score, mask_out);
    TOC3D_LAUNCH_CHECK("toc3d_score_tokens");
    return TOC3D_OK;
}

int toc3d_gumbel_noise(float* out, int64_t n, uint64_t seed, uint64_t* state, toc3d_stream_t stream) {
    TOC3D_REQUIRE(out && state && n >= 0, "toc3d_gumbel_noise: bad arguments");
    TOC3D_REQUIRE(((uintptr_t)out % 16) == 0 && ((uintptr_t)state % 8) == 0, "toc3d_gumbel_noise: out must be 16-byte, state 8-byte aligned");
    if (n == 0) return TOC3D_OK;
    const int64_t blocks = (n + 1023) / 1024;
    TOC3D_REQUIRE(blocks <= 0x7fffffff, "toc3d_gumbel_noise: n too large");
    toc3d_launch(gumbel_noise_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), out, n, (unsigned long long)seed, (unsigned long long*)state);
    TOC3D_LAUNCH_CHECK("toc3d_gumbel_noise");
    return TOC3D_OK;
}

int toc3d_gumbel_from_bits(const uint32_t* bits, int64_t n, float* out, toc3d_stream_t stream) {
    TOC3D_REQUIRE(bits && out && n >= 0, "toc3d_gumbel_from_bits: bad arguments");
    if (n == 0) return TOC3D_OK;
    toc3d_launch(gumbel_from_bits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), bits, n, out);
    TOC3D_LAUNCH_CHECK("toc3d_gumbel_from_bits");
    return TOC3D_OK;
}

int toc3d_score_head(int dtype, const void* f, int64_t ld, int64_t kdim, const float* w, const float* b, const float* gumbel,
                     int64_t M, float* pred, float* score, float* mask_out, toc3d_stream_t stream) {
    TOC3D_REQUIRE(f && w && b && pred && score && mask_out, "toc3d_score_head: null buffer");
    TOC3D_REQUIRE(kdim % 8 == 0 && ld >= kdim && ld % 8 == 0, "toc3d_score_head: kdim / ld must be multiples of 8");
    if (M <= 0) return TOC3D_OK;
    dim3 grid((unsigned)((M + 3) / 4));
    if (dtype == TOC3D_BF16)
        toc3d_launch(score_head_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)f, ld, (int)kdim, w, b, gumbel, M, pred, score, mask_out);
    else if (dtype == TOC3D_F32)
        toc3d_launch(score_head_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)f, ld, (int)kdim, w, b, gumbel, M, pred, score, mask_out);
    else { toc3d_set_error("toc3d_score_head: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_score_head");
    return TOC3D_OK;
}

int toc3d_global_mean_half(int dtype, void* t, int64_t ld, int64_t V, int64_t T, int64_t C, toc3d_stream_t stream) {
    TOC3D_REQUIRE(t && C % 2 == 0 && ld >= C, "toc3d_global_mean_half: bad arguments");
    if (V <= 0 || T <= 0) return TOC3D_OK;
    dim3 grid((unsigned)((C / 2 + 63) / 64), (unsigned)V);
    if (dtype == TOC3D_BF16) toc3d_launch(global_mean_half_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (bf16_t*)t, ld, (int)T, (int)C);
    else if (dtype == TOC3D_F32) toc3d_launch(global_mean_half_kernel<float>, grid, dim3(256), 0, as_stream(stream), (float*)t, ld, (int)T, (int)C);
    else { toc3d_set_error("toc3d_global_mean_half: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_global_mean_half");
    return TOC3D_OK;
}

int toc3d_abs_pos_bicubic(const float* pos, int64_t S, int64_t C, float* out, int64_t h, int64_t w, toc3d_stream_t stream) {
    TOC3D_REQUIRE(pos && out && S > 0 && C > 0 && h > 0 && w > 0, "toc3d_abs_pos_bicubic: bad arguments");
    if (S == h && S == w) {
        hipError_t e = hipMemcpyAsync(out, pos, (size_t)S * S * C * 4, hipMemcpyDeviceToDevice, as_stream(stream));
        if (e != hipSuccess) { toc3d_set_error("toc3d_abs_pos_bicubic: copy failed: %s", hipGetErrorString(e)); return TOC3D_ERR_LAUNCH; }
        return TOC3D_OK;
    }
    toc3d_launch(abs_pos_bicubic_kernel, dim3((unsigned)(h * w)), dim3(256), 0, as_stream(stream), pos, (int)S, (int)C, out, (int)h, (int)w);
    TOC3D_LAUNCH_CHECK("toc3d_abs_pos_bicubic");
    return TOC3D_OK;
}

int toc3d_im2col_3x3(int dtype, const float* x, void* out, int64_t ldo, int64_t V, int64_t h, int64_t w, int64_t C,
                     toc3d_stream_t stream) {
    TOC3D_REQUIRE(x && out && ldo >= 9 * C, "toc3d_im2col_3x3: bad arguments");
    const int64_t M = V * h * w;
    if (M <= 0) return TOC3D_OK;
    if (dtype == TOC3D_BF16) toc3d_launch(im2col_3x3_kernel<bf16_t>, dim3((unsigned)M), dim3(256), 0, as_stream(stream), x, (bf16_t*)out, ldo, (int)V, (int)h, (int)w, (int)C);
    else if (dtype == TOC3D_F32) toc3d_launch(im2col_3x3_kernel<float>, dim3((unsigned)M), dim3(256), 0, as_stream(stream), x, (float*)out, ldo, (int)V, (int)h, (int)w, (int)C);
    else { toc3d_set_error("toc3d_im2col_3x3: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_im2col_3x3");
    return TOC3D_OK;
}

}  // extern "C"
