// Token-level (HBM-bound) kernels of the ToC3D backbone: LayerNorm, per-window top-k ranking with index
// compaction in LDS, gather + merge + LayerNorm of the kept set, scatter back (gfx950).
//
// Reference: Block.norm1/norm2 (backbones/eva_vit.py:249,263; toc3d_eva_vit.py:372,382), SwiGLU.ffn_ln
// (eva_vit.py:48), ScoreBasedTokenSelector.sample (backbones/toc3d_utils.py:131-143), window_partition with
// pad -1e6 (toc3d_eva_vit.py:412-415), batch_index_select / merge_tokens / batch_index_fill
// (toc3d_utils.py:28-70), fast-token update (toc3d_eva_vit.py:449-467).
//
// All arithmetic is f32; rows are moved with 16-byte accesses, one wavefront (64 lanes) per token row.
#include "capi.h"
#include "common.h"

// Row blocks -> XCDs (round 5).  Workgroup b of a launch runs on XCD b % 8; the GEMMs give every XCD a contiguous band of row tiles (xcd_remap / order 1), and what a producer
// leaves in its XCD's L2 is worth 7-8 % of the frame (profiles/r05_nt_stores.txt).  With plain `row = 4 b + wave` a row kernel's rows are dealt round robin over the XCDs: its
// reads of a GEMM's output and its writes of the next GEMM's A operand all cross XCDs.  Bits of TOC3D_ROW_BANDS put a kernel's row blocks in the same bands instead
// (0: LayerNorm rows / rebase, 1: gather's kept rows, 2: scatter): +0.4 % / -0.1 % / +0.3 % frames/s, 1 | 4 shipped (profiles/r05_row_bands.txt).  Speed only.
#ifndef TOC3D_ROW_BANDS
#define TOC3D_ROW_BANDS 5
#endif
#define TOC3D_ROW_BLOCK_IF(bit, b, n) ((TOC3D_ROW_BANDS >> (bit)) & 1 ? xcd_remap((int)(b), (int)(n)) : (int)(b))
#define TOC3D_ROW_BLOCK_OFF_IF(bit, b, off, n) ((TOC3D_ROW_BANDS >> (bit)) & 1 ? xcd_remap_off((int)(b), (int)(off), (int)(n)) : (int)(b) - (int)(off))

namespace {

constexpr float PAD_SCORE = -1.0e6f;        // toc3d_eva_vit.py:415

// ---------------------------------------------------------------------------------------------------
// LayerNorm of one row held by one wavefront.  v[i] = float4 #(lane + 64 i) of the row (nvec valid).
// Two-pass statistics (mean, then centred variance), biased variance, like torch.nn.LayerNorm.
// ---------------------------------------------------------------------------------------------------
template <int MAXV>
TOC3D_DEV void wave_ln_stats(const f32x4 (&v)[MAXV], int nvec, int lane, int C, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nvec) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nvec) {
            const float d0 = v[i][0] - mean, d1 = v[i][1] - mean, d2 = v[i][2] - mean, d3 = v[i][3] - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
}

static bool planes_rows_ok(const void* p, int64_t ld) { return ((uintptr_t)p % 128) == 0 && ld % 32 == 0; }

template <typename T> TOC3D_DEV void store4(T* p, f32x4 v);
template <> TOC3D_DEV void store4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> TOC3D_DEV void store4<f32p_t>(f32p_t* p, f32x4 v) { const float x[4] = {v[0], v[1], v[2], v[3]}; store4_planes(p, x); }
template <> TOC3D_DEV void store4<bf16_t>(bf16_t* p, f32x4 v) {
    bf16x4 b;
    b[0] = (bf16_t)v[0]; b[1] = (bf16_t)v[1]; b[2] = (bf16_t)v[2]; b[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4*>(p) = b;
}

template <typename T, int MAXV>
TOC3D_DEV void wave_ln_write(const f32x4 (&v)[MAXV], int nvec, int lane, float mean, float rstd, const float* __restrict__ gamma,
                             const float* __restrict__ beta, T* __restrict__ out) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < nvec) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * vi);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 4 * vi);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * gm[e] + bt[e];
            store4<T>(out + 4 * vi, y);
        }
    }
}

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ row_index,
                                                      const float* __restrict__ row_scale, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, T* __restrict__ out, int64_t ldo,
                                                      int M, int C) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = TOC3D_ROW_BLOCK_IF(0, blockIdx.x, gridDim.x) * 4 + wave;
    if (row >= M) return;
    const int src = row_index ? row_index[row] : row;
    const float sc = (row_scale && src >= 0) ? row_scale[src] : 1.f;
    const int nvec = C >> 2;
    f32x4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (vi < nvec && src >= 0) {
            v[i] = *reinterpret_cast<const f32x4*>(x + (int64_t)src * ldx + 4 * vi);
            if (row_scale) v[i] *= sc;
        }
    }
    float mean, rstd;
    wave_ln_stats<MAXV>(v, nvec, lane, C, eps, mean, rstd);
    wave_ln_write<T, MAXV>(v, nvec, lane, mean, rstd, gamma, beta, out + (int64_t)row * ldo);
}

// LayerNorm over an act row with n valid columns of ld (ffn_ln over the SwiGLU hidden): chunks of 8 elements,
// one wavefront per row, row cached in registers, gamma/beta read as 16-byte vectors.
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void ln_act_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, T* __restrict__ out, int64_t ldo,
                                                     int M, int n) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    const int nch = (int)(ldo >> 3);
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ci = lane + 64 * i;
        if (ci < nch && ci * 8 < n) {
            load8(x + (int64_t)row * ldx + ci * 8, v[i]);
            if (ci * 8 + 8 > n) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (ci * 8 + e >= n) v[i][e] = 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
        s += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
    }
    const float mean = wave_sum(s) / (float)n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ci = lane + 64 * i;
        float qq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; qq += d * d; }
        if (ci * 8 + 8 <= n) q += qq;
        else if (ci * 8 < n) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (ci * 8 + e < n) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)n + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ci = lane + 64 * i;
        if (ci >= nch) continue;
        float y[8];
        if (ci * 8 + 8 <= n) {
            float gm[8], bt[8];
            load8(gamma + ci * 8, gm);
            load8(beta + ci * 8, bt);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = (v[i][e] - mean) * rstd * gm[e] + bt[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = ci * 8 + e;
                y[e] = c < n ? (v[i][e] - mean) * rstd * gamma[c] + beta[c] : 0.f;
            }
        }
        store8(out + (int64_t)row * ldo + ci * 8, y);
    }
}

// ---------------------------------------------------------------------------------------------------
// Ranking.  rank_j = #{i : s_i > s_j  or (s_i == s_j and i < j)}  == position of j in a stable descending sort.
// ---------------------------------------------------------------------------------------------------
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// One 64-bit key per slot whose unsigned order is the stable descending order: the score's bits made monotone
// (-0 folded onto +0 so equal scores tie), then the complemented slot so the earlier slot wins a tie.
TOC3D_DEV unsigned long long sort_key(float f, int i) {
    unsigned int u = __float_as_uint(f + 0.0f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)i);
}

// number of keys in [lo, hi) that sort before kj; lo even, keys 16-byte aligned (one compare + one add per key)
TOC3D_DEV int count_before(const unsigned long long* keys, int lo, int hi, unsigned long long kj) {
    int r = 0, i = lo;
    for (; i + 8 <= hi; i += 8) {
        u64x2 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const u64x2*>(keys + i + 2 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) r += (q[u].x > kj ? 1 : 0) + (q[u].y > kj ? 1 : 0);
    }
    for (; i < hi; ++i) r += keys[i] > kj ? 1 : 0;
    return r;
}

__global__ __launch_bounds__(256) void rank_desc_kernel(const float* __restrict__ scores, int n, int64_t* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_key[];
    const int b = blockIdx.y;
    const float* sc = scores + (int64_t)b * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s_key[i] = sort_key(sc[i], i);
    __syncthreads();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    order[(int64_t)b * n + count_before(s_key, 0, n, s_key[j])] = j;
}

// number of real (non-padded) slots of window `win` of side L over an h x w token grid
TOC3D_DEV int window_real_count(int win, int h, int w, int L) {
    const int nWh = (h + L - 1) / L, nWw = (w + L - 1) / L;
    const int wr = (win / nWw) % nWh, wc = win % nWw;
    return min(L, h - wr * L) * min(L, w - wc * L);
}

// Per-window selection + compaction (all in LDS).  The kept ("slow") set of a window is its k best slots; kept
// *padded* slots are all the same row (x = 0 -> LN(0) = beta) whose block output is thrown away by
// window_unpartition, so they get no row in the compact buffers: window i owns cap_i = min(k, real_i) + 1 compact
// rows [off_i, off_i + cap_i) = kept real tokens (sorted order), then -- only if a real token ever lost against a pad,
// which needs a log-prob <= -1e6 -- explicit zero rows, then the representative token.  The remaining kept pads are
// *virtual attention keys* (arows = -1, RoPE slot in aslots): cap_i + virtual_i = k + 1 keys, cap_i queries.
// 1024 threads rank the window (up to eight threads share a slot's sweep over the keys); the first 256 then do the
// bookkeeping, in the reduction orders the compact layout was validated with.
__global__ __launch_bounds__(1024) void window_topk_kernel(const float* __restrict__ scores, int V, int h, int w, int L, int k,
                                                           int32_t* __restrict__ order, int32_t* __restrict__ tok, float* __restrict__ wgt,
                                                           int32_t* __restrict__ prow, int32_t* __restrict__ crow_tok,
                                                           int32_t* __restrict__ rep_index, int32_t* __restrict__ rep_row,
                                                           int32_t* __restrict__ arows, int32_t* __restrict__ aslots,
                                                           int32_t* __restrict__ acount_q, int32_t* __restrict__ acount_k,
                                                           int32_t* __restrict__ crow_rc) {
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    constexpr int WORKERS = 256;
    const int N = L * L;
    const int Ne = (N + 1) & ~1;
    unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_raw);   // [N] sort key by slot
    float* s_sc = reinterpret_cast<float*>(s_key + Ne);          // [N] score by slot
    int32_t* s_pref = reinterpret_cast<int32_t*>(s_sc + N);      // [N] rank by slot, then real tokens ranked before rank p
    int32_t* s_ord = s_pref + N;                                 // [N] slot by rank
    float* s_red = reinterpret_cast<float*>(s_ord + N);          // [4] wave partials
    int32_t* s_ired = reinterpret_cast<int32_t*>(s_red + 4);     // [8] int partials + [4] scan partials
    const int nWh = (h + L - 1) / L, nWw = (w + L - 1) / L;
    const int win = blockIdx.x;
    const int v = win / (nWh * nWw), wr = (win / nWw) % nWh, wc = win % nWw;
    const int tid = threadIdx.x;
    const bool worker = tid < WORKERS;
    // token row of a window slot, -1 for a padded slot
    auto slot_tok = [&](int j) -> int {
        const int r = wr * L + j / L, c = wc * L + j % L;
        return (r < h && c < w) ? (v * h + r) * w + c : -1;
    };
    for (int j = tid; j < N; j += blockDim.x) {
        const int t = slot_tok(j);
        const float sc = t >= 0 ? scores[t] : PAD_SCORE;
        s_sc[j] = sc;
        s_key[j] = sort_key(sc, j);
        s_pref[j] = 0;
    }
    // compact-row offset of this window: sum of the (static) capacities of the windows before it
    int offp = 0;
    if (worker)
        for (int i = tid; i < win; i += WORKERS) offp += min(k, window_real_count(i, h, w, L)) + 1;
    __syncthreads();
    // rank of slot j = number of slots that sort before it (descending score, ties by slot): `parts` threads per
    // slot each count over one slice of the keys and add into the slot's rank (integer LDS atomics)
    {
        const int parts = max(1, min(8, (int)blockDim.x / N));
        const int chunk = (((N + parts - 1) / parts) + 1) & ~1;
        for (int id = tid; id < N * parts; id += blockDim.x) {
            const int j = id % N, part = id / N;
            const int lo = part * chunk, hi = min(N, lo + chunk);
            if (lo < hi) atomicAdd(&s_pref[j], count_before(s_key, lo, hi, s_key[j]));
        }
    }
    __syncthreads();
    for (int j = tid; j < N; j += blockDim.x) s_ord[s_pref[j]] = j;
    __syncthreads();
    // denominator of merge_tokens: sum of the fast (dropped) scores, pads included (toc3d_utils.py:68);
    // number of real tokens among the kept; window offset
    float part = 0.f;
    if (worker)
        for (int p = k + tid; p < N; p += WORKERS) part += s_sc[s_ord[p]];
    // exclusive prefix count of real tokens by rank (block scan over the workers, 256 ranks per round)
    int nreal = 0;
    {
        const int lane = tid & 63, wave = tid >> 6;
        int carry = 0;
        for (int base = 0; base < N; base += WORKERS) {
            const int p = base + tid;
            const int f = (worker && p < N && slot_tok(s_ord[p]) >= 0) ? 1 : 0;
            if (p < k) nreal += f;
            int x = f;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
            if (worker && lane == 63) s_ired[8 + wave] = x;
            __syncthreads();
            if (worker) {
                int woff = 0;
                for (int i = 0; i < wave; ++i) woff += s_ired[8 + i];
                const int tot = s_ired[8] + s_ired[9] + s_ired[10] + s_ired[11];
                if (p < N) s_pref[p] = carry + woff + x - f;
                carry += tot;
            }
            __syncthreads();
        }
    }
    part = wave_sum(part);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nreal += __shfl_xor(nreal, o, 64); offp += __shfl_xor(offp, o, 64); }
    if (worker && (tid & 63) == 0) { s_red[tid >> 6] = part; s_ired[tid >> 6] = nreal; s_ired[4 + (tid >> 6)] = offp; }
    __syncthreads();
    if (!worker) return;
    const float denom = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    const int r_w = s_ired[0] + s_ired[1] + s_ired[2] + s_ired[3];
    const int off = s_ired[4] + s_ired[5] + s_ired[6] + s_ired[7];
    const int cap = min(k, window_real_count(win, h, w, L)) + 1;
    const int e_w = cap - 1 - r_w;                               // explicit zero rows (0 unless a real token scored <= -1e6)
    const int kk = k + 1;
    for (int p = tid; p < N; p += WORKERS) {
        const int slot = s_ord[p];
        const int t = slot_tok(slot);
        order[(int64_t)win * N + p] = slot;
        tok[(int64_t)win * N + p] = t;
        wgt[(int64_t)win * N + p] = p >= k ? s_sc[slot] / denom : 0.f;
        int pr = -1;
        if (p < k) {
            const int nr = s_pref[p];                            // real tokens ranked before p
            int j;                                               // position in the window's key list
            if (t >= 0) j = nr;
            else { const int q = p - nr; j = q < e_w ? r_w + q : cap + (q - e_w); }
            if (j < cap - 1) {                                   // explicit compact row
                pr = off + j;
                crow_tok[pr] = t;
                rep_index[pr] = -1;
                if (crow_rc) crow_rc[pr] = ((slot / L) << 16) | (slot % L);     // RoPE position of the compact row (toc3d_linear_qkv_rope)
            }
            arows[(int64_t)win * kk + j] = pr;
            aslots[(int64_t)win * kk + j] = slot;
        }
        prow[(int64_t)win * N + p] = pr;
    }
    if (tid == 0) {
        const int rr = off + cap - 1;                            // representative token: last row of the window, RoPE slot k
        crow_tok[rr] = -2;
        rep_index[rr] = win;
        rep_row[win] = rr;
        if (crow_rc) crow_rc[rr] = ((k / L) << 16) | (k % L);
        arows[(int64_t)win * kk + cap - 1] = rr;
        aslots[(int64_t)win * kk + cap - 1] = k;                 // toc3d_eva_vit.py:434
        acount_q[win] = cap;
        acount_k[win] = kk;
    }
}

// ---------------------------------------------------------------------------------------------------
// gather + merge + LN1.  1024-thread workgroups (16 waves):
//   blocks [0, nW)           : representative token of window i (only when k < N): 16 waves stride over the
//                              fast set, LDS tree (fixed order -> deterministic), wave 0 normalises it;
//   blocks [nW, nW + ceil(nW*k/16)) : one wave per kept row: copy + LayerNorm.
// ---------------------------------------------------------------------------------------------------
// The PREVIOUS block's scatter folded into this gather (toc3d_gather_merge_ln_pending): a token's current value is not in x yet -- it is the
// previous selection's compact row (kept tokens) or x + the previous representative token's branch outputs (dropped tokens), exactly what
// toc3d_scatter_update would have written.  inv[token] >= 0: compact row of the previous selection; < 0: -(1 + previous window).  The wave
// that gathers a token (every real token is gathered exactly once: kept -> copied, dropped -> merged) also writes that value back to x.
struct PendingScatter {
    const int32_t* inv; const float* slow; const float* r1; const float* r2; const float* r3; const float* r4; float* xw;
};

template <bool PENDING>
TOC3D_DEV f32x4 token_value(const float* __restrict__ x, int C, int src, int pk, int vi, const PendingScatter& pd) {
    if constexpr (!PENDING) {
        return *reinterpret_cast<const f32x4*>(x + (int64_t)src * C + 4 * vi);
    } else {
        f32x4 v;
        if (pk >= 0) {
            v = *reinterpret_cast<const f32x4*>(pd.slow + (int64_t)pk * C + 4 * vi);
        } else {
            const int64_t wo = (int64_t)(-1 - pk) * C + 4 * vi;
            v = *reinterpret_cast<const f32x4*>(pd.xw + (int64_t)src * C + 4 * vi);      // (through the writable alias: x itself is declared __restrict__)
            v = (v + *reinterpret_cast<const f32x4*>(pd.r1 + wo)) + *reinterpret_cast<const f32x4*>(pd.r2 + wo);       // toc3d_eva_vit.py:454-456, the scatter kernel's order
            if (pd.r3) v = (v + *reinterpret_cast<const f32x4*>(pd.r3 + wo)) + *reinterpret_cast<const f32x4*>(pd.r4 + wo);
        }
        *reinterpret_cast<f32x4*>(pd.xw + (int64_t)src * C + 4 * vi) = v;
        return v;
    }
}

template <typename T, int MAXV, bool PENDING>
__global__ __launch_bounds__(1024) void gather_merge_ln_kernel(const float* __restrict__ x, int C, const int32_t* __restrict__ tok,
                                                               const float* __restrict__ wgt, const int32_t* __restrict__ crow_tok,
                                                               const int32_t* __restrict__ rep_row, int nW, int N, int k, int Ms,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                               float* __restrict__ shortcut, T* __restrict__ a_out, int64_t lda, int kept_copy,
                                                               PendingScatter pd) {
    extern __shared__ __attribute__((aligned(16))) float s_part[];        // [16][C]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nvec = C >> 2;
    const int rep_blocks = nW;
    if ((int)blockIdx.x < rep_blocks) {
        const int win = blockIdx.x;
        f32x4 acc[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // this wave owns fast positions p = k + wave + 16 j; lane j prefetches index/weight j, then rows are
        // fetched four at a time so the tok -> x dependency is paid once per four rows, not per row
        const int mine = (N - k - wave + 15) / 16;
        int my_src = -1, my_pk = 0;
        float my_w = 0.f;
        if (lane < mine) {
            my_src = tok[(int64_t)win * N + k + wave + 16 * lane];
            my_w = wgt[(int64_t)win * N + k + wave + 16 * lane];
            if (PENDING && my_src >= 0) my_pk = pd.inv[my_src];
        }
        for (int j0 = 0; j0 < mine; j0 += 4) {
            int src[4], pk[4];
            float wg[4];
            f32x4 row[4][MAXV];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // (v_readlane: the lane index is wave-uniform; __shfl would be a ds_bpermute round trip between the index load and the row loads it feeds)
                src[u] = __builtin_amdgcn_readlane(my_src, (j0 + u) & 63);
                pk[u] = PENDING ? __builtin_amdgcn_readlane(my_pk, (j0 + u) & 63) : 0;
                wg[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_w), (j0 + u) & 63));
                if (j0 + u >= mine) src[u] = -1;         // padded slots (src < 0) contribute x = 0; their weight is in the denominator
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int vi = lane + 64 * i;
                    row[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (vi < nvec && src[u] >= 0) row[u][i] = token_value<PENDING>(x, C, src[u], pk[u], vi, pd);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < MAXV; ++i) acc[i] += wg[u] * row[u][i];
        }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) *reinterpret_cast<f32x4*>(s_part + (int64_t)wave * C + 4 * vi) = acc[i];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 1024) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < 16; ++wv) s += s_part[wv * C + c];
            s_part[c] = s;                               // column c of partial 0 is only read by this thread
        }
        __syncthreads();
        if (wave != 0) return;
        f32x4 v[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = lane + 64 * i;
            v[i] = vi < nvec ? *reinterpret_cast<const f32x4*>(s_part + 4 * vi) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int64_t orow = rep_row[win];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) *reinterpret_cast<f32x4*>(shortcut + orow * C + 4 * vi) = v[i];
        }
        float mean, rstd;
        wave_ln_stats<MAXV>(v, nvec, lane, C, eps, mean, rstd);
        wave_ln_write<T, MAXV>(v, nvec, lane, mean, rstd, gamma, beta, a_out + orow * lda);
        return;
    }
    const int64_t orow = (int64_t)TOC3D_ROW_BLOCK_OFF_IF(1, blockIdx.x, rep_blocks, gridDim.x - rep_blocks) * 16 + wave;
    if (orow >= Ms) return;
    const int src = crow_tok[orow];
    if (src == -2) return;                               // representative row: written by its window's block above
    const int pk = (PENDING && src >= 0) ? pd.inv[src] : 0;
    f32x4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (vi < nvec) {
            if (src >= 0) v[i] = token_value<PENDING>(x, C, src, pk, vi, pd);
            // the f32 copy of a kept row is only needed when the projection GEMM reads its residual from the compact buffer; with
            // kept_copy == 0 that GEMM gathers the row from x itself (toc3d_linear_fused residual_index) and 40 % of this kernel's bytes go
            // (explicit zero rows, src < 0, keep theirs: the GEMM reads rows without a token in place)
            if (kept_copy || src < 0) *reinterpret_cast<f32x4*>(shortcut + orow * C + 4 * vi) = v[i];
        }
    }
    float mean, rstd;
    wave_ln_stats<MAXV>(v, nvec, lane, C, eps, mean, rstd);
    wave_ln_write<T, MAXV>(v, nvec, lane, mean, rstd, gamma, beta, a_out + orow * lda);
}


// ---------------------------------------------------------------------------------------------------
// gather + merge + LN1, SPLIT form (round 4).  The single-workgroup merge above is bound by what ONE CU can ingest: a window's dropped rows are
// up to 128-280 x 4 KB = 0.5-1.1 MB through one CU's ~60 GB/s load path (8-17 us) while the 200+ one-row workgroups finish in 2-3 us.  Here a window's
// merge is cut over GM_SPLIT (2 ... 16, default 4) workgroups of 16 / GM_SPLIT wavefronts; wavefront w of slice q IS wavefront (16 / GM_SPLIT) q + w of the kernel above (same rows k + wv + 16 j,
// same order of accumulation), leaves its partial in scratch[win][wv][C], and the slice that arrives last (one agent-scope release per slice, one
// relaxed ticket, one acquire by the last: cdna_hip_programming.md Guideline 16, counter form) adds the 16 partials in the order wv = 0..15 --
// the fixed tree of the kernel above -- and normalises: BIT-IDENTICAL to it (tests/test_gpu_ops.py).  The slices of a window are placed on one XCD
// (blocks b, b + 8, b + 16, ...; speed only).  counters: one word per window, zero before the first launch; the last slice re-arms its word.
// ---------------------------------------------------------------------------------------------------
template <typename T, int MAXV, int GM_SPLIT>
__global__ __launch_bounds__(1024 / GM_SPLIT) void gather_merge_ln_split_kernel(const float* __restrict__ x, int C, const int32_t* __restrict__ tok,
                                                                     const float* __restrict__ wgt, const int32_t* __restrict__ crow_tok,
                                                                     const int32_t* __restrict__ rep_row, int nW, int N, int k, int Ms,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                     float* __restrict__ shortcut, T* __restrict__ a_out, int64_t lda, int kept_copy,
                                                                     float* __restrict__ partials, unsigned* __restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) float s_row[];         // [C] + the ticket
    constexpr int WPB = 16 / GM_SPLIT, NTHR = 64 * WPB;                   // wavefronts / threads per workgroup
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nvec = C >> 2;
    const int rep_blocks = ((nW + 7) / 8) * 8 * GM_SPLIT;
    if ((int)blockIdx.x < rep_blocks) {
        const int b = blockIdx.x;
        const int win = (b / (8 * GM_SPLIT)) * 8 + (b & 7), q = (b >> 3) % GM_SPLIT;
        if (win >= nW) return;
        const int wv = WPB * q + wave;                   // the wavefront of the single-workgroup kernel this one stands for
        f32x4 acc[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int mine = (N - k - wv + 15) / 16;
        int my_src = -1;
        float my_w = 0.f;
        if (lane < mine) {
            my_src = tok[(int64_t)win * N + k + wv + 16 * lane];
            my_w = wgt[(int64_t)win * N + k + wv + 16 * lane];
        }
        constexpr int U = 8;                             // rows in flight per wavefront (the kernel above: 4; the order of the additions is the same)
        for (int j0 = 0; j0 < mine; j0 += U) {
            int src[U];
            float wg[U];
            f32x4 row[U][MAXV];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                src[u] = __builtin_amdgcn_readlane(my_src, (j0 + u) & 63);
                wg[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_w), (j0 + u) & 63));
                if (j0 + u >= mine) { src[u] = -1; wg[u] = 0.f; }
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int vi = lane + 64 * i;
                    row[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (vi < nvec && src[u] >= 0) row[u][i] = *reinterpret_cast<const f32x4*>(x + (int64_t)src[u] * C + 4 * vi);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (j0 + u < mine) {                     // (wave-uniform) rows past the end add nothing -- not even + 0 * 0, which would turn a -0 sum into +0
#pragma unroll
                    for (int i = 0; i < MAXV; ++i) acc[i] += wg[u] * row[u][i];
                }
        }
        float* part = partials + ((int64_t)win * 16 + wv) * C;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) *reinterpret_cast<f32x4*>(part + 4 * vi) = acc[i];
        }
        // publish: every wave's stores acknowledged -> workgroup barrier -> one lane releases at agent scope and takes the window's ticket
        unsigned* s_ticket = reinterpret_cast<unsigned*>(s_row + C);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the compiler may drop the wait behind buffer_wbl2: restated where it cannot)
            *s_ticket = __hip_atomic_fetch_add(counters + win, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (*s_ticket != (unsigned)(GM_SPLIT - 1)) return;
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(counters + win, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch (a recorded plan needs no memset)
        }
        __syncthreads();
        const float* pw = partials + (int64_t)win * 16 * C;
        for (int vi = threadIdx.x; vi < nvec; vi += NTHR) {
            f32x4 p16[16];
#pragma unroll
            for (int w2 = 0; w2 < 16; ++w2) p16[w2] = *reinterpret_cast<const f32x4*>(pw + (int64_t)w2 * C + 4 * vi);
            f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w2 = 0; w2 < 16; ++w2) sum += p16[w2];           // wv = 0..15 in sequence: the fixed tree of gather_merge_ln_kernel
            *reinterpret_cast<f32x4*>(s_row + 4 * vi) = sum;
        }
        __syncthreads();
        if (wave != 0) return;
        f32x4 v[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = lane + 64 * i;
            v[i] = vi < nvec ? *reinterpret_cast<const f32x4*>(s_row + 4 * vi) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int64_t orow = rep_row[win];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) *reinterpret_cast<f32x4*>(shortcut + orow * C + 4 * vi) = v[i];
        }
        float mean, rstd;
        wave_ln_stats<MAXV>(v, nvec, lane, C, eps, mean, rstd);
        wave_ln_write<T, MAXV>(v, nvec, lane, mean, rstd, gamma, beta, a_out + orow * lda);
        return;
    }
    const int64_t orow = (int64_t)TOC3D_ROW_BLOCK_OFF_IF(1, blockIdx.x, rep_blocks, gridDim.x - rep_blocks) * WPB + wave;
    if (orow >= Ms) return;
    const int src = crow_tok[orow];
    if (src == -2) return;                               // representative row: written by its window's last slice above
    f32x4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (vi < nvec) {
            if (src >= 0) v[i] = *reinterpret_cast<const f32x4*>(x + (int64_t)src * C + 4 * vi);
            if (kept_copy || src < 0) *reinterpret_cast<f32x4*>(shortcut + orow * C + 4 * vi) = v[i];
        }
    }
    float mean, rstd;
    wave_ln_stats<MAXV>(v, nvec, lane, C, eps, mean, rstd);
    wave_ln_write<T, MAXV>(v, nvec, lane, mean, rstd, gamma, beta, a_out + orow * lda);
}

// token -> slot of a selection: inv[token] = compact row (kept) or -(1 + window) (dropped); pad slots (tok < 0) have no token.
__global__ __launch_bounds__(256) void token_inverse_map_kernel(const int32_t* __restrict__ tok, const int32_t* __restrict__ prow, int nW, int N, int k,
                                                                int32_t* __restrict__ inv) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)nW * N) return;
    const int t = tok[id];
    if (t < 0) return;
    const int p = (int)(id % N);
    inv[t] = p < k ? prow[id] : -1 - (int)(id / N);
}

// scatter the slow rows back and add the representative token's branch outputs to the fast rows, in place.
__global__ __launch_bounds__(256) void scatter_update_kernel(float* __restrict__ x, int C, const int32_t* __restrict__ tok,
                                                             const int32_t* __restrict__ prow, int nW, int N, int k,
                                                             const float* __restrict__ slow_out, const float* __restrict__ r1,
                                                             const float* __restrict__ r2, const float* __restrict__ r3,
                                                             const float* __restrict__ r4) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t id = (int64_t)TOC3D_ROW_BLOCK_IF(2, blockIdx.x, gridDim.x) * 4 + wave;
    if (id >= (int64_t)nW * N) return;
    const int win = (int)(id / N), p = (int)(id % N);
    const int dst = tok[id];
    if (dst < 0) return;
    const int nvec = C >> 2;
    float* xr = x + (int64_t)dst * C;
    if (p < k) {
        const float* s = slow_out + (int64_t)prow[id] * C;
        for (int vi = lane; vi < nvec; vi += 64) *reinterpret_cast<f32x4*>(xr + 4 * vi) = *reinterpret_cast<const f32x4*>(s + 4 * vi);
    } else {
        const float* a = r1 + (int64_t)win * C;
        const float* b = r2 + (int64_t)win * C;
        for (int vi = lane; vi < nvec; vi += 64) {
            f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * vi);
            v = (v + *reinterpret_cast<const f32x4*>(a + 4 * vi)) + *reinterpret_cast<const f32x4*>(b + 4 * vi);   // toc3d_eva_vit.py:454-456
            if (r3) {                                // a second block on the same kept set: its two updates, in the reference's order
                v = (v + *reinterpret_cast<const f32x4*>(r3 + (int64_t)win * C + 4 * vi)) + *reinterpret_cast<const f32x4*>(r4 + (int64_t)win * C + 4 * vi);
            }
            *reinterpret_cast<f32x4*>(xr + 4 * vi) = v;
        }
    }
}

// [V, T, C] -> [V, C, T] through a 32x32 LDS tile (materialised permute(0,3,1,2), toc3d_eva_vit.py:294)
// Carried compact set (backbone carry_compact): norm1 of a block that continues on the previous block's compact rows.  The block
// would re-merge its representative token from the updated dropped tokens,  sum_j w_j (x_j + delta) = rep_in + W * delta  with W =
// sum of the merge weights of the window's REAL dropped tokens (1 for a full window; ~0 for a ragged one, whose -1e6 pad scores
// swallow the normalisation, toc3d_utils.py:65-70).  The previous block left rep_in + delta in its compact row: representative rows
// first subtract (1 - W) * delta, delta = rep1 + rep2 (written back to the f32 rows), then every row is normalised as usual.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void ln_rebase_kernel(float* __restrict__ slow, int C, const int32_t* __restrict__ rep_index,
                                                        const int32_t* __restrict__ tok, const float* __restrict__ wgt, int N, int k,
                                                        const float* __restrict__ r1, const float* __restrict__ r2,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        T* __restrict__ out, int64_t ldo, int M) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = TOC3D_ROW_BLOCK_IF(0, blockIdx.x, gridDim.x) * 4 + wave;
    if (row >= M) return;
    const int nvec = C >> 2;
    const int win = rep_index[row];                      // wave-uniform
    float* xr = slow + (int64_t)row * C;
    f32x4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + 64 * i;
        v[i] = vi < nvec ? *reinterpret_cast<const f32x4*>(xr + 4 * vi) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (win >= 0) {
        float w = 0.f;
        for (int j = k + lane; j < N; j += 64)
            if (tok[(int64_t)win * N + j] >= 0) w += wgt[(int64_t)win * N + j];
        const float f = 1.0f - wave_sum(w);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(r1 + (int64_t)win * C + 4 * vi) + *reinterpret_cast<const f32x4*>(r2 + (int64_t)win * C + 4 * vi);
                v[i] -= f * d;
                *reinterpret_cast<f32x4*>(xr + 4 * vi) = v[i];
            }
        }
    }
    float mean, rstd;
    wave_ln_stats<MAXV>(v, nvec, lane, C, eps, mean, rstd);
    wave_ln_write<T, MAXV>(v, nvec, lane, mean, rstd, gamma, beta, out + (int64_t)row * ldo);
}

// plain device-to-device copy as a kernel, so that it can live inside a recorded launch plan (16-byte body, byte tail)
__global__ __launch_bounds__(256) void copy_bytes_kernel(char* __restrict__ dst, const char* __restrict__ src, int64_t n16, int64_t nbytes) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // (round 5: four loads in flight per lane before the first store read 4.3 TB/s where this plain loop reads 5.0-5.2: the grid already keeps 8 MB in flight)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[i];
    for (int64_t j = n16 * 16 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nbytes; j += stride) dst[j] = src[j];
}

// Pull a read-only buffer (the next block's packed weights) towards the chip ahead of its use: plain 16-byte loads whose values are
// kept alive but never stored.  The lines land in the Infinity Cache (and the L2 of the XCD the workgroup runs on); the GEMM that follows
// then starts on cache hits instead of HBM misses.  Few workgroups on purpose: this runs beside the block chain on a side lane.
constexpr int PREFETCH_MAX_SEGS = 8;
struct PrefetchSegs { const f32x4* ptr[PREFETCH_MAX_SEGS]; int64_t n16[PREFETCH_MAX_SEGS]; };

__global__ __launch_bounds__(256) void prefetch_kernel(PrefetchSegs sg, float* __restrict__ sink) {
    const f32x4* __restrict__ p = sg.ptr[blockIdx.y];
    const int64_t n16 = sg.n16[blockIdx.y];
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {              // four independent loads in flight per thread
        const f32x4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += (a + b) + (c + d);
    }
    for (; i < n16; i += stride) acc += p[i];
    // never true for finite data that is not all-NaN; keeps the loads observable without writing anything
    if (acc[0] != acc[0] && acc[1] != acc[1] && acc[2] != acc[2] && acc[3] != acc[3] && sink) sink[0] = acc[0];
}

// several small device-to-device copies in ONE launch (the per-frame inputs of the recorded plan: every launch has a ~5 us floor)
constexpr int COPY_MAX_SEGS = 16;
struct CopySegs { char* dst[COPY_MAX_SEGS]; const char* src[COPY_MAX_SEGS]; int64_t nbytes[COPY_MAX_SEGS]; };

__global__ __launch_bounds__(256) void copy_segments_kernel(CopySegs c) {
    const int sg = blockIdx.y;
    char* dst = c.dst[sg];
    const char* src = c.src[sg];
    const int64_t nbytes = c.nbytes[sg];
    const bool al = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
    const int64_t n16 = al ? nbytes / 16 : 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const f32x4* __restrict__ s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(dst);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {              // four independent 16-byte loads in flight per lane before the first store (round 5: 5.0 -> see profiles/r05)
        const f32x4 a = s4[i], b = s4[i + stride], c = s4[i + 2 * stride], e = s4[i + 3 * stride];
        d4[i] = a; d4[i + stride] = b; d4[i + 2 * stride] = c; d4[i + 3 * stride] = e;
    }
    for (; i < n16; i += stride) d4[i] = s4[i];
    for (int64_t j = n16 * 16 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nbytes; j += stride) dst[j] = src[j];
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int C) {
    __shared__ float tile[32][33];
    const int v = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        tile[i][tx] = (t < T && c < C) ? x[((int64_t)v * T + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        if (t < T && c < C) out[((int64_t)v * C + c) * T + t] = tile[tx][i];
    }
}

}  // namespace

extern "C" {

int toc3d_layernorm_rows(int dtype, const float* x, int64_t ldx, const int32_t* row_index, const float* row_scale,
                         const float* gamma, const float* beta, float eps, void* out, int64_t ldo, int64_t M, int64_t C,
                         toc3d_stream_t stream) {
    TOC3D_REQUIRE(x && gamma && beta && out, "toc3d_layernorm_rows: null buffer");
    TOC3D_REQUIRE(C > 0 && C % 4 == 0 && C <= 2048, "toc3d_layernorm_rows: C=%lld must be a multiple of 4 and <= 2048", (long long)C);
    TOC3D_REQUIRE(ldx >= C && ldo >= C && ldx % 4 == 0 && ldo % 4 == 0, "toc3d_layernorm_rows: bad leading dims");
    if (M <= 0) return TOC3D_OK;
    dim3 grid((unsigned)((M + 3) / 4)), block(256);
    hipStream_t s = as_stream(stream);
#define LNR(T, MV) toc3d_launch((ln_rows_kernel<T, MV>), grid, block, 0, s, x, ldx, row_index, row_scale, gamma, beta, eps, (T*)out, ldo, (int)M, (int)C)
    if (dtype == TOC3D_BF16) { if (C <= 1024) LNR(bf16_t, 4); else LNR(bf16_t, 8); }
    else if (dtype == TOC3D_F32) { if (C <= 1024) LNR(float, 4); else LNR(float, 8); }
    else if (dtype == TOC3D_F32X3P) {            // f32 arithmetic, rows written as (hi, lo) planes: the A operand of a bf16 x 3 GEMM
        TOC3D_REQUIRE(planes_rows_ok(out, ldo), "toc3d_layernorm_rows: rows of (hi, lo) planes start on 128-byte boundaries (out aligned, ldo a multiple of 32)");
        if (C <= 1024) LNR(f32p_t, 4); else LNR(f32p_t, 8);
    }
    else { toc3d_set_error("toc3d_layernorm_rows: bad dtype"); return TOC3D_ERR_ARG; }
#undef LNR
    TOC3D_LAUNCH_CHECK("toc3d_layernorm_rows");
    return TOC3D_OK;
}

int toc3d_layernorm_act(int dtype, const void* x, int64_t ldx, const float* gamma, const float* beta, float eps, void* out,
                        int64_t ldo, int64_t M, int64_t n, toc3d_stream_t stream) {
    TOC3D_REQUIRE(x && gamma && beta && out, "toc3d_layernorm_act: null buffer");
    TOC3D_REQUIRE(n > 0 && ldo % 8 == 0 && ldx % 8 == 0 && ldo >= n && ldx >= ldo, "toc3d_layernorm_act: need n <= ldo <= ldx, multiples of 8");
    TOC3D_REQUIRE(ldo <= 64 * 8 * 12, "toc3d_layernorm_act: row too long (%lld > 6144)", (long long)ldo);
    if (M <= 0) return TOC3D_OK;
    dim3 grid((unsigned)((M + 3) / 4)), block(256);
    hipStream_t s = as_stream(stream);
#define LNA(T, MC) toc3d_launch((ln_act_kernel<T, MC>), grid, block, 0, s, (const T*)x, ldx, gamma, beta, eps, (T*)out, ldo, (int)M, (int)n)
    const int mc = (int)((ldo / 8 + 63) / 64);
    if (dtype == TOC3D_BF16) { if (mc <= 2) LNA(bf16_t, 2); else if (mc <= 6) LNA(bf16_t, 6); else LNA(bf16_t, 12); }
    else if (dtype == TOC3D_F32) { if (mc <= 2) LNA(float, 2); else if (mc <= 6) LNA(float, 6); else LNA(float, 12); }
    else { toc3d_set_error("toc3d_layernorm_act: bad dtype"); return TOC3D_ERR_ARG; }
#undef LNA
    TOC3D_LAUNCH_CHECK("toc3d_layernorm_act");
    return TOC3D_OK;
}

int toc3d_rank_desc(const float* scores, int64_t B, int64_t n, int64_t* order, toc3d_stream_t stream) {
    TOC3D_REQUIRE(scores && order, "toc3d_rank_desc: null buffer");
    TOC3D_REQUIRE(n >= 0 && n <= 16000, "toc3d_rank_desc: n=%lld exceeds the LDS-resident limit 16000", (long long)n);
    if (B <= 0 || n == 0) return TOC3D_OK;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    const size_t lds = (size_t)n * 8;
    static Toc3dLdsAttr rank_attr;
    if (lds > 64 * 1024) rank_attr.ensure(reinterpret_cast<const void*>(&rank_desc_kernel), 160 * 1024);
    toc3d_launch(rank_desc_kernel, grid, dim3(256), lds, as_stream(stream), scores, (int)n, order);
    TOC3D_LAUNCH_CHECK("toc3d_rank_desc");
    return TOC3D_OK;
}

int64_t toc3d_window_topk_rows(int64_t V, int64_t h, int64_t w, int64_t L, int64_t k) {
    if (V <= 0 || h <= 0 || w <= 0 || L <= 0 || k < 0) return -1;
    const int64_t nWh = (h + L - 1) / L, nWw = (w + L - 1) / L;
    int64_t rows = 0;
    for (int64_t wr = 0; wr < nWh; ++wr)
        for (int64_t wc = 0; wc < nWw; ++wc) {
            const int64_t real = (L < h - wr * L ? L : h - wr * L) * (L < w - wc * L ? L : w - wc * L);
            rows += (k < real ? k : real) + 1;
        }
    return rows * V;
}

int toc3d_window_topk(const float* scores, int64_t V, int64_t h, int64_t w, int64_t L, int64_t k, int32_t* order,
                      int32_t* tok, float* wgt, int32_t* prow, int32_t* crow_tok, int32_t* rep_index, int32_t* rep_row,
                      int32_t* arows, int32_t* aslots, int32_t* acount_q, int32_t* acount_k, int32_t* crow_rc, toc3d_stream_t stream) {
    TOC3D_REQUIRE(scores && order && tok && wgt && prow && crow_tok && rep_index && rep_row && arows && aslots && acount_q && acount_k,
                  "toc3d_window_topk: null buffer");
    TOC3D_REQUIRE(V > 0 && h > 0 && w > 0 && L > 0 && L <= 64, "toc3d_window_topk: bad dims");
    const int64_t N = L * L;
    TOC3D_REQUIRE(k >= 0 && k < N, "toc3d_window_topk: k=%lld outside [0, %lld) (keep-all is the dense Block path)", (long long)k, (long long)N);
    const int nW = (int)(V * ((h + L - 1) / L) * ((w + L - 1) / L));
    const size_t lds = (size_t)((N + 1) & ~(int64_t)1) * 8 + (size_t)N * 12 + 64;
    static Toc3dLdsAttr topk_attr;
    if (lds > 64 * 1024) topk_attr.ensure(reinterpret_cast<const void*>(&window_topk_kernel), 96 * 1024);
    toc3d_launch(window_topk_kernel, dim3(nW), dim3(1024), lds, as_stream(stream), scores, (int)V, (int)h, (int)w,
                       (int)L, (int)k, order, tok, wgt, prow, crow_tok, rep_index, rep_row, arows, aslots, acount_q, acount_k, crow_rc);
    TOC3D_LAUNCH_CHECK("toc3d_window_topk");
    return TOC3D_OK;
}

}  // extern "C"

static int launch_gather(int dtype, const float* x, int64_t C, const int32_t* tok, const float* wgt, const int32_t* crow_tok, const int32_t* rep_row,
                         int64_t nW, int64_t N, int64_t k, int64_t rows, const float* gamma, const float* beta, float eps, float* shortcut, void* a_out,
                         int64_t lda, int64_t kept_copy, PendingScatter pd, toc3d_stream_t stream) {
    dim3 grid((unsigned)(nW + (rows + 15) / 16)), block(1024);
    const size_t lds = (size_t)16 * C * 4;
    hipStream_t s = as_stream(stream);
#define TOC3D_GATHER(T, P)                                                                                                          \
    do {                                                                                                                            \
        static Toc3dLdsAttr attr;                                                                                                   \
        attr.ensure(reinterpret_cast<const void*>(&gather_merge_ln_kernel<T, 4, P>), 65536);                                        \
        toc3d_launch((gather_merge_ln_kernel<T, 4, P>), grid, block, lds, s, x, (int)C, tok, wgt, crow_tok, rep_row, (int)nW, (int)N, (int)k, (int)rows, \
                     gamma, beta, eps, shortcut, (T*)a_out, lda, (int)(kept_copy != 0), pd);                                        \
    } while (0)
    if (dtype == TOC3D_BF16) TOC3D_GATHER(bf16_t, false);
    else if (dtype == TOC3D_F32) TOC3D_GATHER(float, false);
    else if (dtype == TOC3D_F32X3P) {            // a_out as (hi, lo) planes (the q|k|v GEMM's A operand); shortcut stays f32
        TOC3D_REQUIRE(planes_rows_ok(a_out, lda), "toc3d_gather_merge_ln: rows of (hi, lo) planes start on 128-byte boundaries (a_out aligned, lda a multiple of 32)");
        TOC3D_GATHER(f32p_t, false);
    }
    else { toc3d_set_error("toc3d_gather_merge_ln: bad dtype"); return TOC3D_ERR_ARG; }
#undef TOC3D_GATHER
    TOC3D_LAUNCH_CHECK("toc3d_gather_merge_ln");
    return TOC3D_OK;
}

extern "C" {

int toc3d_gather_merge_ln_ex(int dtype, const float* x, int64_t C, const int32_t* tok, const float* wgt, const int32_t* crow_tok,
                             const int32_t* rep_row, int64_t nW, int64_t N, int64_t k, int64_t rows, const float* gamma,
                             const float* beta, float eps, float* shortcut, void* a_out, int64_t lda, int64_t kept_copy, toc3d_stream_t stream) {
    TOC3D_REQUIRE(x && tok && wgt && crow_tok && rep_row && gamma && beta && shortcut && a_out, "toc3d_gather_merge_ln: null buffer");
    TOC3D_REQUIRE(C > 0 && C % 4 == 0 && C <= 1024, "toc3d_gather_merge_ln: C=%lld must be a multiple of 4 and <= 1024", (long long)C);
    TOC3D_REQUIRE(k >= 0 && k < N && lda >= C && lda % 4 == 0 && rows >= nW, "toc3d_gather_merge_ln: bad k / lda / rows");
    // each of the 16 waves prefetches the index / weight of its share of the dropped tokens in its 64 lanes
    TOC3D_REQUIRE(N - k <= 1024, "toc3d_gather_merge_ln: N - k = %lld dropped tokens per window exceed the kernel's 1024", (long long)(N - k));
    if (nW <= 0) return TOC3D_OK;
    return launch_gather(dtype, x, C, tok, wgt, crow_tok, rep_row, nW, N, k, rows, gamma, beta, eps, shortcut, a_out, lda, kept_copy, PendingScatter{}, stream);
}


// scratch = [TOC3D_GATHER_SPLIT_COUNTER_BYTES of arrival counters][nW][16][C] f32 partials.  The counter region has a FIXED size: launches of different nW
// (the two window types of a plan: 126 and 60 windows at 640x1600, 96 and 36 at B = 2) may share one scratch buffer on one stream -- with a launch-dependent
// start of the partials, the float data of the small-nW launch overwrote counter words of the large-nW one.
int64_t toc3d_gather_merge_ln_scratch_bytes(int64_t nW, int64_t C) {
    if (nW <= 0 || C <= 0) return 0;
    return TOC3D_GATHER_SPLIT_COUNTER_BYTES + nW * 16 * C * 4;
}

int toc3d_gather_merge_ln_split(int dtype, const float* x, int64_t C, const int32_t* tok, const float* wgt, const int32_t* crow_tok,
                                const int32_t* rep_row, int64_t nW, int64_t N, int64_t k, int64_t rows, const float* gamma,
                                const float* beta, float eps, float* shortcut, void* a_out, int64_t lda, int64_t kept_copy,
                                void* scratch, int64_t scratch_bytes, int64_t split, toc3d_stream_t stream) {
    TOC3D_REQUIRE(split == 0 || split == 2 || split == 4 || split == 8 || split == 16, "toc3d_gather_merge_ln_split: split = 2, 4, 8, 16 workgroups per window (0 = the default, 4)");
    TOC3D_REQUIRE(x && tok && wgt && crow_tok && rep_row && gamma && beta && shortcut && a_out, "toc3d_gather_merge_ln_split: null buffer");
    TOC3D_REQUIRE(C > 0 && C % 4 == 0 && C <= 1024, "toc3d_gather_merge_ln_split: C=%lld must be a multiple of 4 and <= 1024", (long long)C);
    TOC3D_REQUIRE(k >= 0 && k < N && lda >= C && lda % 4 == 0 && rows >= nW && N - k <= 1024, "toc3d_gather_merge_ln_split: bad k / lda / rows");
    TOC3D_REQUIRE(scratch && ((uintptr_t)scratch % 256) == 0 && scratch_bytes >= toc3d_gather_merge_ln_scratch_bytes(nW, C),
                  "toc3d_gather_merge_ln_split: scratch of toc3d_gather_merge_ln_scratch_bytes(nW, C) bytes, 256-byte aligned, zeroed once by the caller");
    TOC3D_REQUIRE(nW * 4 <= TOC3D_GATHER_SPLIT_COUNTER_BYTES, "toc3d_gather_merge_ln_split: at most %d windows per launch", TOC3D_GATHER_SPLIT_COUNTER_BYTES / 4);
    if (nW <= 0) return TOC3D_OK;
    unsigned* counters = reinterpret_cast<unsigned*>(scratch);
    float* partials = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + TOC3D_GATHER_SPLIT_COUNTER_BYTES);
    const int sp = split == 0 ? 4 : (int)split, wpb = 16 / sp;
    dim3 grid((unsigned)(((nW + 7) / 8) * 8 * sp + (rows + wpb - 1) / wpb)), block(64 * wpb);
    const size_t lds = (size_t)C * 4 + 16;
    hipStream_t s = as_stream(stream);
#define TOC3D_GSPLIT(T, SP)                                                                                                                       \
    toc3d_launch((gather_merge_ln_split_kernel<T, 4, SP>), grid, block, lds, s, x, (int)C, tok, wgt, crow_tok, rep_row, (int)nW, (int)N, (int)k, (int)rows, \
                 gamma, beta, eps, shortcut, (T*)a_out, lda, (int)(kept_copy != 0), partials, counters)
#define TOC3D_GSPLIT_T(T)                                                                                            \
    do { if (sp == 2) TOC3D_GSPLIT(T, 2); else if (sp == 4) TOC3D_GSPLIT(T, 4); else if (sp == 8) TOC3D_GSPLIT(T, 8); else TOC3D_GSPLIT(T, 16); } while (0)
    if (dtype == TOC3D_BF16) TOC3D_GSPLIT_T(bf16_t);
    else if (dtype == TOC3D_F32) TOC3D_GSPLIT_T(float);
    else if (dtype == TOC3D_F32X3P) {
        TOC3D_REQUIRE(planes_rows_ok(a_out, lda), "toc3d_gather_merge_ln_split: rows of (hi, lo) planes start on 128-byte boundaries (a_out aligned, lda a multiple of 32)");
        TOC3D_GSPLIT_T(f32p_t);
    }
    else { toc3d_set_error("toc3d_gather_merge_ln_split: bad dtype"); return TOC3D_ERR_ARG; }
#undef TOC3D_GSPLIT_T
#undef TOC3D_GSPLIT
    TOC3D_LAUNCH_CHECK("toc3d_gather_merge_ln_split");
    return TOC3D_OK;
}


int toc3d_gather_merge_ln(int dtype, const float* x, int64_t C, const int32_t* tok, const float* wgt, const int32_t* crow_tok,
                          const int32_t* rep_row, int64_t nW, int64_t N, int64_t k, int64_t rows, const float* gamma,
                          const float* beta, float eps, float* shortcut, void* a_out, int64_t lda, toc3d_stream_t stream) {
    return toc3d_gather_merge_ln_ex(dtype, x, C, tok, wgt, crow_tok, rep_row, nW, N, k, rows, gamma, beta, eps, shortcut, a_out, lda, 1, stream);
}

int toc3d_scatter_update(float* x, int64_t C, const int32_t* tok, const int32_t* prow, int64_t nW, int64_t N, int64_t k,
                         const float* slow_out, const float* rep_raw1, const float* rep_raw2, const float* rep_raw3, const float* rep_raw4,
                         toc3d_stream_t stream) {
    TOC3D_REQUIRE(x && tok && prow && slow_out && rep_raw1 && rep_raw2, "toc3d_scatter_update: null buffer");
    TOC3D_REQUIRE((rep_raw3 == nullptr) == (rep_raw4 == nullptr), "toc3d_scatter_update: rep_raw3 and rep_raw4 come as a pair");
    TOC3D_REQUIRE(C > 0 && C % 4 == 0 && k >= 0 && k < N, "toc3d_scatter_update: bad dims");
    if (nW <= 0 || N <= 0) return TOC3D_OK;
    dim3 grid((unsigned)((nW * N + 3) / 4));
    toc3d_launch(scatter_update_kernel, grid, dim3(256), 0, as_stream(stream), x, (int)C, tok, prow, (int)nW, (int)N, (int)k, slow_out, rep_raw1, rep_raw2, rep_raw3, rep_raw4);
    TOC3D_LAUNCH_CHECK("toc3d_scatter_update");
    return TOC3D_OK;
}

int toc3d_rebase_layernorm_rows(int dtype, float* slow, int64_t C, const int32_t* rep_index, const int32_t* tok, const float* wgt, int64_t N,
                                int64_t k, const float* rep_raw1, const float* rep_raw2, const float* gamma, const float* beta, float eps,
                                void* out, int64_t ldo, int64_t rows, toc3d_stream_t stream) {
    TOC3D_REQUIRE(slow && rep_index && tok && wgt && rep_raw1 && rep_raw2 && gamma && beta && out, "toc3d_rebase_layernorm_rows: null buffer");
    TOC3D_REQUIRE(C > 0 && C % 4 == 0 && C <= 2048 && ldo >= C && ldo % 4 == 0 && k >= 0 && k < N, "toc3d_rebase_layernorm_rows: bad dims");
    if (rows <= 0) return TOC3D_OK;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t s = as_stream(stream);
#define LNB(T, MV) toc3d_launch((ln_rebase_kernel<T, MV>), grid, block, 0, s, slow, (int)C, rep_index, tok, wgt, (int)N, (int)k, rep_raw1, rep_raw2, gamma, beta, eps, (T*)out, ldo, (int)rows)
    if (dtype == TOC3D_BF16) { if (C <= 1024) LNB(bf16_t, 4); else LNB(bf16_t, 8); }
    else if (dtype == TOC3D_F32) { if (C <= 1024) LNB(float, 4); else LNB(float, 8); }
    else if (dtype == TOC3D_F32X3P) {
        TOC3D_REQUIRE(planes_rows_ok(out, ldo), "toc3d_rebase_layernorm_rows: rows of (hi, lo) planes start on 128-byte boundaries (out aligned, ldo a multiple of 32)");
        if (C <= 1024) LNB(f32p_t, 4); else LNB(f32p_t, 8);
    }
    else { toc3d_set_error("toc3d_rebase_layernorm_rows: bad dtype"); return TOC3D_ERR_ARG; }
#undef LNB
    TOC3D_LAUNCH_CHECK("toc3d_rebase_layernorm_rows");
    return TOC3D_OK;
}

int toc3d_copy_bytes(void* dst, const void* src, int64_t nbytes, toc3d_stream_t stream) {
    TOC3D_REQUIRE(nbytes >= 0 && (nbytes == 0 || (dst && src)), "toc3d_copy_bytes: bad arguments");
    if (nbytes == 0) return TOC3D_OK;
    const bool aligned = ((uintptr_t)dst % 16) == 0 && ((uintptr_t)src % 16) == 0;
    const int64_t n16 = aligned ? nbytes / 16 : 0;
    const int64_t work = n16 > 0 ? n16 : nbytes;
    const int blocks = (int)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
    toc3d_launch(copy_bytes_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (char*)dst, (const char*)src, n16, nbytes);
    TOC3D_LAUNCH_CHECK("toc3d_copy_bytes");
    return TOC3D_OK;
}


int toc3d_copy_segments(int64_t n, void* const* dst, const void* const* src, const int64_t* nbytes, toc3d_stream_t stream) {
    TOC3D_REQUIRE(n >= 0 && n <= COPY_MAX_SEGS && (n == 0 || (dst && src && nbytes)), "toc3d_copy_segments: 0 <= n <= %d segments, host arrays of n entries", COPY_MAX_SEGS);
    if (n == 0) return TOC3D_OK;
    CopySegs c;
    int64_t most = 0;
    for (int64_t i = 0; i < n; ++i) {
        TOC3D_REQUIRE(nbytes[i] >= 0 && (nbytes[i] == 0 || (dst[i] && src[i])), "toc3d_copy_segments: bad segment %lld", (long long)i);
        c.dst[i] = (char*)dst[i]; c.src[i] = (const char*)src[i]; c.nbytes[i] = nbytes[i];
        most = nbytes[i] > most ? nbytes[i] : most;
    }
    if (most == 0) return TOC3D_OK;
    const int64_t per = (most / 16 + 255) / 256;
    const int bx = (int)(per < 1 ? 1 : (per > 64 ? 64 : per));
    toc3d_launch(copy_segments_kernel, dim3(bx, (unsigned)n), dim3(256), 0, as_stream(stream), c);
    TOC3D_LAUNCH_CHECK("toc3d_copy_segments");
    return TOC3D_OK;
}

int toc3d_nhwc_to_nchw(const float* x, float* out, int64_t V, int64_t T, int64_t C, toc3d_stream_t stream) {
    TOC3D_REQUIRE(x && out && V > 0 && T > 0 && C > 0, "toc3d_nhwc_to_nchw: bad arguments");
    dim3 grid((unsigned)((T + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)V);
    toc3d_launch(nhwc_to_nchw_kernel, grid, dim3(256), 0, as_stream(stream), x, out, (int)T, (int)C);
    TOC3D_LAUNCH_CHECK("toc3d_nhwc_to_nchw");
    return TOC3D_OK;
}

}  // extern "C"
