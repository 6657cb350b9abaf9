// Windowed multi-head self-attention with RoPE-by-slot for the EVA-02 blocks (gfx950).
//
// Reference: backbones/eva_vit.py:101-113 (dense windows), backbones/toc3d_eva_vit.py:499-512 (kept tokens,
// RoPE rows gathered by slot index, backbones/eva_utils.py:396-403), rotate_half eva_utils.py:318-322.
//
// One workgroup (4 waves) = one (window, head, 64-query tile); wave w owns 16 query rows.  Keys/values of the
// window stream through LDS in tiles of 64 (K row-major with RoPE already applied, V transposed so the P.V
// B-fragment is 8 consecutive keys), online softmax in f32 registers; P goes C-layout -> LDS -> A-fragment
// inside its own wave (no workgroup barrier).  The window never exceeds 400 + 1 keys, so K/V re-reads by the
// window's other query tiles are L2 hits.
//
// Dense blocks zero-pad *after* LayerNorm (eva_vit.py:249-254): a padded slot is a key with k = 0 (k_proj has
// no bias, eva_vit.py:98) and v = v_bias, identical for every pad, so the npad virtual keys are folded in
// analytically at the end: denominator += npad * exp(0 - m), numerator += npad * exp(0 - m) * v_bias.
#include "capi.h"
#include "common.h"

namespace {

constexpr int HD = 64;           // head dim (EVA-02 L: 1024 / 16)
constexpr int KT = 64;           // keys per LDS tile
constexpr float NEG_BIG = -1.0e30f;

struct AttnArgs {
    const void* qkv; int64_t ldqkv;
    void* out; int64_t ldo;
    const int32_t* rows; const int32_t* slots; const int32_t* count; const int32_t* count_k; const int32_t* npad;
    const void* pad_qkv;
    int64_t stride;
    int C;
    const float* cosT; const float* sinT; const float* v_bias;
    float scale;
};

template <typename T> struct Pad;                // LDS row padding (elements) keeping 16-byte alignment
template <> struct Pad<bf16_t> { static constexpr int ld = HD + 8; };
template <> struct Pad<float> { static constexpr int ld = HD + 4; };

// rotate 8 consecutive head-dim elements (4 pairs) by the table row; eva_utils.py:379:
//   out[2t] = x[2t]*cos[2t] - x[2t+1]*sin[2t],  out[2t+1] = x[2t+1]*cos[2t+1] + x[2t]*sin[2t+1]
TOC3D_DEV void rope8(float (&x)[8], const float* __restrict__ c, const float* __restrict__ s) {
    float cs[8], sn[8];
    load8(c, cs);
    load8(s, sn);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float a = x[2 * t], b = x[2 * t + 1];
        x[2 * t] = a * cs[2 * t] - b * sn[2 * t];
        x[2 * t + 1] = b * cs[2 * t + 1] + a * sn[2 * t + 1];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs a) {
    constexpr int LD = Pad<T>::ld;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ks = reinterpret_cast<T*>(smem);          // [KT keys][LD]
    T* Vt = Ks + KT * LD;                        // [HD dims][LD]   (keys along the row)
    T* Ps = Vt + HD * LD;                        // [4 waves][16 q][LD]

    const int qt = blockIdx.x, head = blockIdx.y, win = blockIdx.z;
    const int n = a.count[win];                  // queries: the window's compact rows
    if (qt * 64 >= n) return;                    // uniform for the workgroup
    const int nkeys = a.count_k ? a.count_k[win] : n;   // keys: the same rows + virtual kept-pad keys (rows[j] < 0)
    const int32_t* rows = a.rows + (int64_t)win * a.stride;
    const int32_t* slots = a.slots + (int64_t)win * a.stride;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* padq = reinterpret_cast<const T*>(a.pad_qkv);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r16 = lane & 15, g = lane >> 4;
    const bool wave_active = qt * 64 + wave * 16 < n;

    // ---- Q fragments (A operand: row = r16, k = g*8 + j + 32*s), RoPE + scale applied in f32 ----
    Frag<T> qf[2];
    {
        const int qi = qt * 64 + wave * 16 + r16;
        const bool ok = qi < n;
        const int qrow = ok ? rows[qi] : 0, qslot = ok ? slots[qi] : 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int d0 = s * 32 + g * 8;
            float x[8];
            load8(qkv + (int64_t)qrow * a.ldqkv + head * HD + d0, x);
            rope8(x, a.cosT + (int64_t)qslot * HD + d0, a.sinT + (int64_t)qslot * HD + d0);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = ok ? x[j] * a.scale : 0.f;
            qf[s] = make_frag(x, T());
        }
    }

    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m[4], l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = NEG_BIG; l[r] = 0.f; }

    T* Pw = Ps + wave * 16 * LD;
    const int nkt = (nkeys + KT - 1) / KT;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                         // previous K/V tile fully consumed
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = tid + it * 256;        // 512 (key, 8-dim chunk) pairs per tile
            const int key = c >> 3, dc = c & 7;
            const int kj = kt * KT + key;
            const bool ok = kj < nkeys;
            const int row = ok ? rows[kj] : 0, slot = ok ? slots[kj] : 0;
            // a kept padded slot is the row LN(0) = beta for every window: its q|k|v is a per-block constant
            const T* src = row >= 0 ? qkv + (int64_t)row * a.ldqkv : padq;
            float kx[8], vx[8];
            load8(src + a.C + head * HD + dc * 8, kx);
            load8(src + 2 * a.C + head * HD + dc * 8, vx);
            rope8(kx, a.cosT + (int64_t)slot * HD + dc * 8, a.sinT + (int64_t)slot * HD + dc * 8);
            if (!ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { kx[j] = 0.f; vx[j] = 0.f; }
            }
            store8(Ks + key * LD + dc * 8, kx);
#pragma unroll
            for (int j = 0; j < 8; ++j) Vt[(dc * 8 + j) * LD + key] = to_act<T>(vx[j]);
        }
        __syncthreads();
        if (!wave_active) continue;              // wave-uniform; barriers stay outside

        // ---- S = Q K^T : 16 q x 64 keys ----
        f32x4 sc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const Frag<T> kf = read_frag(Ks + (t * 16 + r16) * LD + s * 32 + g * 8);
                mma_step(sc[t], qf[s], kf);
            }
        }
        // lane holds S[q = g*4 + r][key = t*16 + r16]; mask keys past the window
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool kok = kt * KT + t * 16 + r16 < nkeys;
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[t][r] = kok ? sc[t][r] : NEG_BIG;
        }
        // ---- online softmax (per q row r; the row lives in the 16 lanes sharing g) ----
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = fmaxf(fmaxf(sc[0][r], sc[1][r]), fmaxf(sc[2][r], sc[3][r]));
            mx = row16_max(mx);
            const float mn = fmaxf(m[r], mx);
            const float alpha = __expf(m[r] - mn);
            float ps = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float p = __expf(sc[t][r] - mn);
                sc[t][r] = p;
                ps += p;
            }
            l[r] = l[r] * alpha + ps;            // per-lane partial; reduced over the 16 lanes at the end
            m[r] = mn;
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d][r] *= alpha;
        }
        // ---- P: C layout -> LDS -> A fragments (own wave only) ----
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Pw[(g * 4 + r) * LD + t * 16 + r16] = to_act<T>(sc[t][r]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- O += P V : A = P[q][key], B = V[key][d] read from Vt[d][key] ----
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const Frag<T> pf = read_frag(Pw + r16 * LD + s * 32 + g * 8);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const Frag<T> vf = read_frag(Vt + (d * 16 + r16) * LD + s * 32 + g * 8);
                mma_step(o[d], pf, vf);
            }
        }
    }
    if (!wave_active) return;

    // ---- epilogue: fold in the virtual zero-padded keys, normalise, store ----
    const int np = a.npad ? a.npad[win] : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float lr = row16_sum(l[r]);
        float alpha = 1.f, padw = 0.f;
        if (np > 0) {
            const float mn = fmaxf(m[r], 0.f);
            alpha = __expf(m[r] - mn);
            padw = (float)np * __expf(-mn);
            lr = lr * alpha + padw;
        }
        const float inv = 1.f / lr;
        const int qi = qt * 64 + wave * 16 + g * 4 + r;
        if (qi < n) {
            const int64_t orow = rows[qi];
            T* dst = reinterpret_cast<T*>(a.out) + orow * a.ldo + head * HD;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int dd = d * 16 + r16;
                float v = o[d][r] * alpha;
                if (np > 0) v += padw * a.v_bias[head * HD + dd];
                dst[dd] = to_act<T>(v * inv);
            }
        }
    }
}

template <typename T>
void launch_attn(const AttnArgs& a, dim3 grid, hipStream_t s) {
    const size_t lds = (size_t)(KT + HD + 4 * 16) * Pad<T>::ld * sizeof(T);
    hipLaunchKernelGGL(attn_kernel<T>, grid, dim3(256), lds, s, a);
}

// window_partition as index maps (backbones/eva_utils.py:89-110): real tokens of each window in slot order.
__global__ void window_map_dense_kernel(int V, int h, int w, int L, int32_t* rows, int32_t* slots, int32_t* count, int32_t* npad) {
    const int nWh = (h + L - 1) / L, nWw = (w + L - 1) / L;
    const int win = blockIdx.x;
    const int v = win / (nWh * nWw), wr = (win / nWw) % nWh, wc = win % nWw;
    const int N = L * L;
    // real region of this window is a rectangle rh x rw anchored at the window's top-left corner
    const int rh = min(L, h - wr * L), rw = min(L, w - wc * L);
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        int32_t row = -1, slot = 0;
        if (j < rh * rw) {
            const int sr = j / rw, sc = j % rw;
            row = (v * h + wr * L + sr) * w + wc * L + sc;
            slot = sr * L + sc;
        }
        rows[(int64_t)win * N + j] = row;
        slots[(int64_t)win * N + j] = slot;
    }
    if (threadIdx.x == 0) { count[win] = rh * rw; npad[win] = N - rh * rw; }
}

}  // namespace

extern "C" {

int toc3d_window_attention(int dtype, const void* qkv, int64_t ldqkv, void* out, int64_t ldo, const int32_t* rows,
                           const int32_t* slots, const int32_t* count, const int32_t* count_k, const int32_t* npad,
                           const void* pad_qkv, int64_t stride, int64_t nwin, int64_t max_count, int64_t num_heads,
                           const float* rope_cos, const float* rope_sin, const float* v_bias, float scale,
                           toc3d_stream_t stream) {
    TOC3D_REQUIRE(dtype == TOC3D_F32 || dtype == TOC3D_BF16, "toc3d_window_attention: bad dtype %d", dtype);
    TOC3D_REQUIRE(qkv && out && rows && slots && count && rope_cos && rope_sin, "toc3d_window_attention: null buffer");
    TOC3D_REQUIRE(!npad || v_bias, "toc3d_window_attention: npad given without v_bias");
    TOC3D_REQUIRE(!count_k || pad_qkv, "toc3d_window_attention: count_k given without pad_qkv");
    TOC3D_REQUIRE(num_heads > 0 && nwin >= 0 && max_count >= 0 && stride >= max_count, "toc3d_window_attention: bad dims");
    const int64_t C = num_heads * HD;
    TOC3D_REQUIRE(ldqkv >= 3 * C && ldo >= C, "toc3d_window_attention: leading dims too small for head_dim 64");
    const int esz = dtype == TOC3D_BF16 ? 2 : 4;
    TOC3D_REQUIRE((ldqkv * esz) % 16 == 0 && ((uintptr_t)qkv % 16) == 0, "toc3d_window_attention: qkv rows must be 16-byte aligned");
    TOC3D_REQUIRE(num_heads <= 65535 && nwin <= 65535, "toc3d_window_attention: grid too large");
    if (nwin == 0 || max_count == 0) return TOC3D_OK;
    AttnArgs a{qkv, ldqkv, out, ldo, rows, slots, count, count_k, npad, pad_qkv, stride, (int)C, rope_cos, rope_sin, v_bias, scale};
    dim3 grid((unsigned)((max_count + 63) / 64), (unsigned)num_heads, (unsigned)nwin);
    if (dtype == TOC3D_BF16) launch_attn<bf16_t>(a, grid, as_stream(stream));
    else launch_attn<float>(a, grid, as_stream(stream));
    TOC3D_LAUNCH_CHECK("toc3d_window_attention");
    return TOC3D_OK;
}

int toc3d_window_map_dense(int64_t V, int64_t h, int64_t w, int64_t L, int32_t* rows, int32_t* slots, int32_t* count,
                           int32_t* npad, toc3d_stream_t stream) {
    TOC3D_REQUIRE(rows && slots && count && npad, "toc3d_window_map_dense: null buffer");
    TOC3D_REQUIRE(V > 0 && h > 0 && w > 0 && L > 0, "toc3d_window_map_dense: bad dims");
    const int nW = (int)(V * ((h + L - 1) / L) * ((w + L - 1) / L));
    hipLaunchKernelGGL(window_map_dense_kernel, dim3(nW), dim3(256), 0, as_stream(stream), (int)V, (int)h, (int)w, (int)L, rows, slots, count, npad);
    TOC3D_LAUNCH_CHECK("toc3d_window_map_dense");
    return TOC3D_OK;
}

}  // extern "C"
