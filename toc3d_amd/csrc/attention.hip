// Windowed multi-head self-attention with RoPE-by-slot for the EVA-02 blocks (gfx950).
//
// Reference: backbones/eva_vit.py:101-113 (dense windows), backbones/toc3d_eva_vit.py:499-512 (kept tokens,
// RoPE rows gathered by slot index, backbones/eva_utils.py:396-403), rotate_half eva_utils.py:318-322.
//
// One workgroup (4 waves) = one (window, head, 64-query tile); wave w owns 16 query rows.  Keys/values of the
// window stream through LDS in tiles of 64 (K row-major with RoPE already applied, V transposed so the P.V
// B-fragment is contiguous keys), online softmax in f32 registers; the scores are computed transposed (S^T = K Q^T)
// so that P stays in registers and feeds P.V directly (see attn_small_kernel).  The window never exceeds 400 + 1
// keys, so K/V re-reads by the window's other query tiles are L2 hits.
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
    int L;                                       // RoPE table side: tables are [L*L, 64]
    const float* cosT; const float* sinT; const float* v_bias;
    float scale;
    // weight prefetch riding on this launch (toc3d_window_attention_pf): the first pf_rows rows of the grid's window dimension stream these
    // buffers through the caches and discard them; nwin = number of real windows (the grid is pf_rows + nwin in its window dimension)
    const f32x4* pf_ptr[4]; int64_t pf_n16[4]; int nwin, pf_rows;
    int out_planes;                              // f32 kernels: the output rows leave as (hi, lo) bf16 planes (TOC3D_DTYPE_F32X3P: the projection GEMM's A operand)
};

// The attention kernels are latency-bound and leave HBM idle, and the GEMMs that follow them start on weights that were last touched a frame
// ago (0.6 GB of bf16 weights cycle through a 256 MB Infinity Cache: 0.22 ms of a 5.7 ms frame, profiles/r02_where_time_goes.txt).  Extra
// workgroups of the attention launch therefore read the weights of the next GEMMs (this block's proj / w1|w2 / w3, the next block's q|k|v) and
// throw the values away -- no extra launch and no cross-stream edge (the standalone prefetch lane lost 3-5 % to exactly those).
TOC3D_DEV void prefetch_weights(const AttnArgs& a, int wg, int nwg, int tid) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int sgm = 0; sgm < 4; ++sgm) {
        const f32x4* __restrict__ p = a.pf_ptr[sgm];
        const int64_t n16 = a.pf_n16[sgm];
        const int64_t stride = (int64_t)nwg * 256;
        int64_t i = (int64_t)wg * 256 + tid;
        for (; i + 3 * stride < n16; i += 4 * stride) {
            const f32x4 x0 = p[i], x1 = p[i + stride], x2 = p[i + 2 * stride], x3 = p[i + 3 * stride];
            acc += (x0 + x1) + (x2 + x3);
        }
        for (; i < n16; i += stride) acc += p[i];
    }
    // never true for finite weights; keeps the loads alive without writing anything anyone reads
    if (acc[0] != acc[0] && acc[1] != acc[1] && acc[2] != acc[2] && acc[3] != acc[3]) reinterpret_cast<volatile float*>(a.out)[0] = 0.f;
}

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

// QM = 16-row MFMA tiles of queries per wave: a workgroup covers 64*QM query rows, so every K/V tile (and its RoPE
// rotation) staged in LDS serves QM times more queries.
TOC3D_DEV void frag_to_float(const Frag<bf16_t>& f, float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)f.v[i];
}
TOC3D_DEV void frag_to_float(const Frag<float>& f, float (&o)[8]) {
    o[0] = f.lo[0]; o[1] = f.lo[1]; o[2] = f.lo[2]; o[3] = f.lo[3]; o[4] = f.hi[0]; o[5] = f.hi[1]; o[6] = f.hi[2]; o[7] = f.hi[3];
}

// one (key, 8-dim chunk) of a K/V tile in flight: raw K, raw V and the RoPE table rows, still in registers
template <typename T> struct KVChunk {
    Frag<T> k, v;
};

// RoPE of 8 consecutive head dims [dc*8, dc*8+8) with the compact axial tables held in LDS.
// VisionRotaryEmbeddingFast (eva_utils.py:362-371): angle[slot][d] = row-part(slot / L) for d < 32, column-part(slot % L)
// for d >= 32, each frequency repeated for the pair (2i, 2i+1) -> tab[part][coord][16 freqs] holds everything.
// rc = (slot / L) << 16 | (slot % L).
TOC3D_DEV void rope8_lds(float (&x)[8], const float* cosRC, const float* sinRC, int L, int rc, int dc) {
    const int part = dc >> 2;
    const int coord = part ? (rc & 0xffff) : (rc >> 16);
    const int off = (part * L + coord) * 16 + (dc & 3) * 4;
    const f32x4 c4 = *reinterpret_cast<const f32x4*>(cosRC + off);
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(sinRC + off);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float xa = x[2 * t], xb = x[2 * t + 1];
        x[2 * t] = xa * c4[t] - xb * s4[t];
        x[2 * t + 1] = xb * c4[t] + xa * s4[t];
    }
}

// V^T row of head dim d.  A (key, 8-dim chunk) thread writes dims dc*8 + j; in natural order the 8 chunks of a wavefront
// instruction land 8 rows apart = on 2 bank groups (measured: SQ_LDS_BANK_CONFLICT 55 % of the LDS-active cycles).  With
// row(d) = (d & 3)*16 + (d >> 2) they land 2 rows apart = 8 distinct bank groups, and the P.V accumulator of MFMA column
// block d', lane column r16 is head dim r16*4 + d': each lane owns 4 consecutive output dims (one 8 / 16-byte store).
TOC3D_DEV int vt_row(int d) { return (d & 3) * 16 + (d >> 2); }

template <typename T> TOC3D_DEV float softmax_exp(float x);
template <> TOC3D_DEV float softmax_exp<bf16_t>(float x) { return __expf(x); }
template <> TOC3D_DEV float softmax_exp<float>(float x) { return expf(x); }      // strict-parity path: the precise routine

TOC3D_DEV Frag<bf16_t> frag_from_halves(const bf16_t* p0, const bf16_t* p1) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p0), b = *reinterpret_cast<const bf16x4*>(p1);
    Frag<bf16_t> f;
    f.v = bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return f;
}
TOC3D_DEV Frag<float> frag_from_halves(const float* p0, const float* p1) {
    Frag<float> f;
    f.lo = *reinterpret_cast<const f32x4*>(p0);
    f.hi = *reinterpret_cast<const f32x4*>(p1);
    return f;
}
TOC3D_DEV bf16_t frag_elem(const Frag<bf16_t>& f, int j) { return f.v[j]; }
TOC3D_DEV float frag_elem(const Frag<float>& f, int j) { return j < 4 ? f.lo[j] : f.hi[j - 4]; }
// (g4_max: common.h)

// ---- bf16 x 3 products inside the f32 kernels (TOC3D_DTYPE_F32X3 / F32X3P: the attention of precision "fp32x3") -------------------------------------
// The exact-f32 form multiplies on v_mfma_f32_16x16x4_f32: 8 instructions (256 cycles) per 16x16x32 step, and these kernels are bound by them (the
// fp32x3 frame spent 14 % of its time here once its GEMMs ran on planes).  X3 keeps every buffer, the RoPE, the softmax and the accumulation in f32 and
// forms the two contractions like the GEMMs of that precision do: a . b = hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16 (48 cycles; relative error of a
// product <= ~2^-16).  Nothing is split inside the MFMA loops: the staged K and V^T tiles keep their f32 LDS image (same addresses, same bank pattern) but
// every aligned 8-element chunk of K [32 bytes] holds [8 x hi | 8 x lo] and every aligned 4-key group of a V^T row [16 bytes] holds [4 x hi | 4 x lo],
// written once by the thread that stages the element; Q is split once per query tile, P once per 32 keys (it feeds four MFMA steps).
struct X3Frag { bf16x8 hi, lo; };
TOC3D_DEV X3Frag x3_split(const float (&x)[8]) {
    X3Frag f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bf16_t h = (bf16_t)x[e];
        f.hi[e] = h;
        f.lo[e] = (bf16_t)(x[e] - (float)h);
    }
    return f;
}
TOC3D_DEV void x3_chunk_bits(const float (&x)[8], float (&bits)[8]) {       // the 32 bytes of a K chunk: [8 x hi | 8 x lo]
    const X3Frag f = x3_split(x);
    const f32x4 a = __builtin_bit_cast(f32x4, f.hi), b = __builtin_bit_cast(f32x4, f.lo);
    bits[0] = a[0]; bits[1] = a[1]; bits[2] = a[2]; bits[3] = a[3]; bits[4] = b[0]; bits[5] = b[1]; bits[6] = b[2]; bits[7] = b[3];
}
TOC3D_DEV X3Frag x3_from_chunk(const Frag<float>& f) { return X3Frag{__builtin_bit_cast(bf16x8, f.lo), __builtin_bit_cast(bf16x8, f.hi)}; }
TOC3D_DEV X3Frag x3_from_halves(const float* p0, const float* p1) {        // two 4-key groups of a V^T row: [4 x hi | 4 x lo] each
    const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(p0)), b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(p1));
    return X3Frag{__builtin_shufflevector(a, b, 0, 1, 2, 3, 8, 9, 10, 11), __builtin_shufflevector(a, b, 4, 5, 6, 7, 12, 13, 14, 15)};
}
TOC3D_DEV void x3_store_vt(float* elem, int key, float v) {                // element (row, key) of V^T; elem = the address the f32 element would have
    const bf16_t h = (bf16_t)v;
    char* grp = reinterpret_cast<char*>(elem - (key & 3));
    *reinterpret_cast<bf16_t*>(grp + 2 * (key & 3)) = h;
    *reinterpret_cast<bf16_t*>(grp + 8 + 2 * (key & 3)) = (bf16_t)(v - (float)h);
}
TOC3D_DEV void x3_mma(f32x4& acc, const X3Frag& a, const X3Frag& b) {      // small terms first, like the GEMM
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, acc, 0, 0, 0);
}

template <typename T, int QM, bool X3 = false>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs a) {
    constexpr int LD = Pad<T>::ld;
    constexpr int QB = 64 * QM;                  // query rows per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ks = reinterpret_cast<T*>(smem);          // [KT keys][LD]
    T* Vt = Ks + KT * LD;                        // [HD dims][LD]   (keys along the row)
    int32_t* s_rows = reinterpret_cast<int32_t*>(Vt + HD * LD);            // [stride] key rows of this window
    const int istride = (int)((a.stride + 3) & ~3);                        // keeps the tables behind the lists 16-byte aligned
    int32_t* s_slots = s_rows + istride;                                   // [stride] RoPE slots as (row << 16 | col)
    float* s_cos = reinterpret_cast<float*>(s_slots + istride);            // [2][L][16] compact axial tables
    float* s_sin = s_cos + 2 * a.L * 16;
    const int L = a.L;

    const int qt = blockIdx.x, head = blockIdx.y, win = (int)blockIdx.z - a.pf_rows;
    if (win < 0) {                               // prefetch rows come FIRST in the grid (dispatched with the first attention workgroups, not behind the last)
        prefetch_weights(a, (blockIdx.z * gridDim.y + head) * gridDim.x + qt, a.pf_rows * gridDim.y * gridDim.x, threadIdx.x);
        return;
    }
    const int n = a.count[win];                  // queries: the window's compact rows
    if (qt * QB >= n) return;                    // uniform for the workgroup
    const int nkeys = a.count_k ? a.count_k[win] : n;   // keys: the same rows + virtual kept-pad keys (rows[j] < 0)
    const int32_t* rows = a.rows + (int64_t)win * a.stride;
    const int32_t* slots = a.slots + (int64_t)win * a.stride;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* padq = reinterpret_cast<const T*>(a.pad_qkv);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r16 = lane & 15, g = lane >> 4;
    const int q0 = qt * QB + wave * 16 * QM;     // first query row of this wave
    const bool wave_active = q0 < n;

    // the window's key list and the compact RoPE tables go to LDS once (5 KB instead of 512 B of table per key per tile)
    for (int j = tid; j < nkeys; j += 256) {
        const int sl = slots[j];
        s_rows[j] = rows[j];
        s_slots[j] = ((sl / L) << 16) | (sl % L);
    }
    for (int i = tid; i < L * 16; i += 256) {
        const int c = i >> 4, f = i & 15;
        s_cos[i] = a.cosT[(int64_t)(c * L) * HD + 2 * f];               // row part: slot (c, 0), dims 0..31
        s_sin[i] = a.sinT[(int64_t)(c * L) * HD + 2 * f];
        s_cos[L * 16 + i] = a.cosT[(int64_t)c * HD + 32 + 2 * f];       // column part: slot (0, c), dims 32..63
        s_sin[L * 16 + i] = a.sinT[(int64_t)c * HD + 32 + 2 * f];
    }
    __syncthreads();
    // ---- Q fragments (A operand: row = r16, k = g*8 + j + 32*s), RoPE + scale applied in f32 ----
    Frag<T> qf[QM][2];
    X3Frag qx[X3 ? QM : 1][2];                   // X3: the same fragments as (hi, lo) pairs (qf unused)
#pragma unroll
    for (int mi = 0; mi < QM; ++mi) {
        const int qi = q0 + mi * 16 + r16;
        const bool ok = qi < n;
        const int qrow = ok ? s_rows[qi] : 0, qrc = ok ? s_slots[qi] : 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int d0 = s * 32 + g * 8;
            float x[8];
            load8(qkv + (int64_t)qrow * a.ldqkv + head * HD + d0, x);
            rope8_lds(x, s_cos, s_sin, L, qrc, d0 >> 3);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = ok ? x[j] * a.scale : 0.f;
            if constexpr (X3) qx[mi][s] = x3_split(x);
            else qf[mi][s] = make_frag(x, T());
        }
    }
    // software pipeline: tile kt+1 is fetched into registers while tile kt is multiplied
    KVChunk<T> pre[2];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = tid + it * 256;        // 512 (key, 8-dim chunk) pairs per tile
            const int key = c >> 3, dc = c & 7;
            const int kj = kt * KT + key;
            const bool ok = kj < nkeys;
            const int row = ok ? s_rows[kj] : 0;
            // a kept padded slot is the row LN(0) = beta for every window: its q|k|v is a per-block constant
            const T* src = row >= 0 ? qkv + (int64_t)row * a.ldqkv : padq;
            pre[it].k = read_frag(src + a.C + head * HD + dc * 8);
            pre[it].v = read_frag(src + 2 * a.C + head * HD + dc * 8);
        }
    };
    fetch(0);

    f32x4 o[QM][4];
    float m[QM], l[QM];                          // running max / partial sum of query r16 (+ mi*16) -- one query per lane, see below
#pragma unroll
    for (int mi = 0; mi < QM; ++mi) {
#pragma unroll
        for (int d = 0; d < 4; ++d) o[mi][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        m[mi] = NEG_BIG;
        l[mi] = 0.f;
    }

    const int nkt = (nkeys + KT - 1) / KT;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                         // previous K/V tile fully consumed
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = tid + it * 256;
            const int key = c >> 3, dc = c & 7;
            T* vdst = Vt + (dc * 2) * LD + key;                  // vt_row(dc*8 + j) = (j & 3)*16 + dc*2 + (j >> 2)
            if (kt * KT + key < nkeys) {
                float kx[8];
                frag_to_float(pre[it].k, kx);
                rope8_lds(kx, s_cos, s_sin, L, s_slots[kt * KT + key], dc);          // rotate_half pairs (eva_utils.py:318-322,379)
                if constexpr (X3) {
                    float kb[8];
                    x3_chunk_bits(kx, kb);
                    store8(Ks + key * LD + dc * 8, kb);
#pragma unroll
                    for (int j = 0; j < 8; ++j) x3_store_vt(reinterpret_cast<float*>(vdst + ((j & 3) * 16 + (j >> 2)) * LD), key, (float)frag_elem(pre[it].v, j));
                } else {
                    store8(Ks + key * LD + dc * 8, kx);
#pragma unroll
                    for (int j = 0; j < 8; ++j) vdst[((j & 3) * 16 + (j >> 2)) * LD] = frag_elem(pre[it].v, j);   // V moves as stored: no conversion
                }
            } else {
                const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                store8(Ks + key * LD + dc * 8, z);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (X3) x3_store_vt(reinterpret_cast<float*>(vdst + ((j & 3) * 16 + (j >> 2)) * LD), key, 0.f);   // (a 4-byte zero would clobber a neighbour's lo half)
                    else vdst[((j & 3) * 16 + (j >> 2)) * LD] = to_act<T>(0.f);
                }
            }
        }
        __syncthreads();
        if (kt + 1 < nkt) fetch(kt + 1);         // in flight during the MFMA phase below
        if (!wave_active) continue;              // wave-uniform; barriers stay outside

        // ---- S^T = K Q^T (K fragment as the A operand): lane holds S[q = r16][key = t*16 + g*4 + r], i.e. scores of ONE query; P then
        // feeds P.V straight from these registers (slot numbering in the comment of attn_small_kernel below) -- no LDS round trip ----
#pragma unroll
        for (int mi = 0; mi < QM; ++mi) {
            f32x4 sc[4];
            float mx = NEG_BIG;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if constexpr (X3) x3_mma(sc[t], x3_from_chunk(read_frag(reinterpret_cast<const float*>(Ks) + (t * 16 + r16) * LD + s * 32 + g * 8)), qx[mi][s]);
                    else mma_step(sc[t], read_frag(Ks + (t * 16 + r16) * LD + s * 32 + g * 8), qf[mi][s]);
                }
                if (kt * KT + t * 16 + 16 > nkeys) {              // mask keys past the window: only a tile that straddles the end (wave-uniform)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[t][r] = kt * KT + t * 16 + g * 4 + r < nkeys ? sc[t][r] : NEG_BIG;
                }
                mx = fmaxf(mx, fmaxf(fmaxf(sc[t][0], sc[t][1]), fmaxf(sc[t][2], sc[t][3])));
            }
            // ---- online softmax of query r16: the 4 lane groups hold disjoint keys of the same query ----
            mx = g4_max(mx);
            const float mn = fmaxf(m[mi], mx);
            const float alpha = softmax_exp<T>(m[mi] - mn);
            float ps = 0.f;
            Frag<T> pf[2];
            X3Frag px[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pv[j] = softmax_exp<T>(sc[2 * c + (j >> 2)][j & 3] - mn);
                    ps += pv[j];
                }
                if constexpr (X3) px[c] = x3_split(pv);
                else pf[c] = make_frag(pv, T());
            }
            l[mi] = l[mi] * alpha + ps;              // per-lane partial; the 4 lane groups are summed at the end
            m[mi] = mn;
            // the accumulator rows are queries g*4 + r: their factor lives in lane g*4 + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float aq = __shfl(alpha, g * 4 + r, 64);
#pragma unroll
                for (int d = 0; d < 4; ++d) o[mi][d][r] *= aq;
            }
            // ---- O += P V : A = P[q][key slot], B = V[key slot][d] read from Vt[d][key] with the same slot numbering ----
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const T* vrow = Vt + (d * 16 + r16) * LD + c * 32 + g * 4;
                    if constexpr (X3) x3_mma(o[mi][d], px[c], x3_from_halves(reinterpret_cast<const float*>(vrow), reinterpret_cast<const float*>(vrow) + 16));
                    else mma_step(o[mi][d], pf[c], frag_from_halves(vrow, vrow + 16));
                }
        }
    }
    if (!wave_active) return;

    // ---- epilogue: fold in the virtual zero-padded keys, normalise, store ----
    const int np = a.npad ? a.npad[win] : 0;
#pragma unroll
    for (int mi = 0; mi < QM; ++mi) {
        float lr = g4_sum(l[mi]);
        float alpha = 1.f, padw = 0.f;
        if (np > 0) {
            const float mn = fmaxf(m[mi], 0.f);
            alpha = softmax_exp<T>(m[mi] - mn);
            padw = (float)np * softmax_exp<T>(-mn);
            lr = lr * alpha + padw;
        }
        const float inv = 1.f / lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float inv_q = __shfl(inv, g * 4 + r, 64), alpha_q = __shfl(alpha, g * 4 + r, 64), padw_q = __shfl(padw, g * 4 + r, 64);
            const int qi = q0 + mi * 16 + g * 4 + r;
            if (qi < n) {
                const int64_t orow = rows[qi];
                T* dst = reinterpret_cast<T*>(a.out) + orow * a.ldo + head * HD + r16 * 4;
                T o4[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {                   // V^T row d*16 + r16 holds head dim r16*4 + d (vt_row)
                    float v = o[mi][d][r] * alpha_q;
                    if (np > 0) v += padw_q * a.v_bias[head * HD + r16 * 4 + d];
                    o4[d] = to_act<T>(v * inv_q);
                }
                if constexpr (sizeof(T) == 4) {
                    if (a.out_planes) store4_planes(dst, o4);
                    else store4(dst, o4);
                } else {
                    store4(dst, o4);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Windows of up to 208 keys (every accelerated block of the shipped configs; dense windows too when they fit): one workgroup per
// (window, head) keeps the whole K (RoPE applied) and V^T of that head in LDS, so nothing is staged twice and the softmax is
// single-pass (no running max / rescale).  Wave w walks the 16-query MFMA tiles w, w+4, ...
//
// The scores are computed TRANSPOSED, S^T = K . Q^T (K fragment as the MFMA A operand, Q as B): a lane then holds scores of ONE
// query (q = lane & 15) for the keys t*16 + (lane >> 4)*4 + 0..3 of every 16-key tile t.  Two tiles give the lane 8 probabilities of
// its query -- exactly an A-operand fragment of P . V if the 32 key slots of that MFMA step are numbered
//     slot (g, j < 4) = key c*32 + g*4 + j,      slot (g, j >= 4) = key c*32 + 16 + g*4 + (j - 4)
// and the V^T fragment is read with the same numbering (two 4-element reads instead of one 8-element read).  P never leaves the
// registers: no LDS round trip of 2-byte scattered stores, no per-tile wave barriers, and no P buffer in LDS (more workgroups per CU).
// Row max / sum live in one lane per query (+ a butterfly over the 4 lane groups); the normalisation of an output row
// q = (lane >> 4)*4 + r fetches 1/sum from lane q.
// Dense blocks (npad): the analytic zero-pad keys of the header comment enter the single pass as max(m, 0), sum += npad * exp(-m).
// ---------------------------------------------------------------------------------------------------
template <typename T, int MAXSUB, bool X3 = false>
__global__ __launch_bounds__(256) void attn_small_kernel(AttnArgs a) {
    constexpr int LD = Pad<T>::ld;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int head = blockIdx.x, win = (int)blockIdx.y - a.pf_rows;
    if (win < 0) {                               // prefetch rows come FIRST in the grid (dispatched with the first attention workgroups, not behind the last)
        prefetch_weights(a, blockIdx.y * gridDim.x + head, a.pf_rows * gridDim.x, threadIdx.x);
        return;
    }
    const int n = a.count[win];
    if (n == 0) return;
    const int nkeys = a.count_k ? a.count_k[win] : n;
    const int nsub = (nkeys + 15) >> 4;          // 16-key MFMA tiles
    const int NK32 = ((nkeys + 31) >> 5) << 5;   // keys padded to the 32-wide P.V step
    const int LDP = NK32 + 16 / (int)sizeof(T);  // row stride of V^T (elements), keeps 16-byte alignment
    T* Ks = reinterpret_cast<T*>(smem);          // [nsub*16][LD]
    T* Vt = Ks + nsub * 16 * LD;                 // [HD][LDP]
    int32_t* s_rows = reinterpret_cast<int32_t*>(Vt + HD * LDP);
    const int istride = (int)((a.stride + 3) & ~3);
    int32_t* s_slots = s_rows + istride;
    float* s_cos = reinterpret_cast<float*>(s_slots + istride);
    float* s_sin = s_cos + 2 * a.L * 16;
    const int L = a.L;
    const int32_t* rows = a.rows + (int64_t)win * a.stride;
    const int32_t* slots = a.slots + (int64_t)win * a.stride;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* padq = reinterpret_cast<const T*>(a.pad_qkv);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r16 = lane & 15, g = lane >> 4;

    // ---- staging with every global load of the workgroup in flight at once (the kernel is latency-bound: its inputs
    // were written by a GEMM on other XCDs, so nothing hits this XCD's L2) ----
    constexpr int MAXC = (MAXSUB * 16 + 16) * 8 / 256;           // (key, 8-dim chunk) pairs per thread
    int crow[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {                             // 1. index loads
        const int key = (tid + i * 256) >> 3;
        crow[i] = key < nkeys ? rows[key] : 0;
    }
    Frag<T> kraw[MAXC], vraw[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {                             // 2. K / V rows (a kept pad = the per-block constant row)
        const int c = tid + i * 256, dc = c & 7;
        if (c < NK32 * 8) {
            const T* src = crow[i] >= 0 ? qkv + (int64_t)crow[i] * a.ldqkv : padq;
            kraw[i] = read_frag(src + a.C + head * HD + dc * 8);
            vraw[i] = read_frag(src + 2 * a.C + head * HD + dc * 8);
        }
    }
    // raw Q rows of all tiles this wave owns are requested in the same round trip (their rows come from the same list)
    const int nmt = (n + 15) >> 4;
    constexpr int MAXT = (MAXSUB + 3) / 4;
    Frag<T> qraw[MAXT][2];
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
        const int qi = (wave + 4 * u) * 16 + r16;
        const int qrow = qi < n ? rows[qi] : -1;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
            if (wave + 4 * u < nmt) {
                const T* src = qrow >= 0 ? qkv + (int64_t)qrow * a.ldqkv : qkv;
                qraw[u][s2] = read_frag(src + head * HD + s2 * 32 + g * 8);
            }
    }
    for (int j = tid; j < nkeys; j += 256) {                     // 3. output rows, RoPE coordinates (one division per key, not per chunk)
        const int sl = slots[j];                                 //    and the compact RoPE tables -> LDS
        s_rows[j] = rows[j];
        s_slots[j] = ((sl / L) << 16) | (sl % L);
    }
    for (int i = tid; i < L * 16; i += 256) {
        const int c = i >> 4, f = i & 15;
        s_cos[i] = a.cosT[(int64_t)(c * L) * HD + 2 * f];
        s_sin[i] = a.sinT[(int64_t)(c * L) * HD + 2 * f];
        s_cos[L * 16 + i] = a.cosT[(int64_t)c * HD + 32 + 2 * f];
        s_sin[L * 16 + i] = a.sinT[(int64_t)c * HD + 32 + 2 * f];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {                             // 4. rotate K, transpose V into LDS; rows past nkeys are zero
        const int c = tid + i * 256;
        if (c < NK32 * 8) {
            const int key = c >> 3, dc = c & 7;
            T* vdst = Vt + (dc * 2) * LDP + key;                 // vt_row(dc*8 + j) = (j & 3)*16 + dc*2 + (j >> 2)
            if (key < nkeys) {
                float kx[8];
                frag_to_float(kraw[i], kx);
                rope8_lds(kx, s_cos, s_sin, L, s_slots[key], dc);
                if constexpr (X3) {
                    float kb[8];
                    x3_chunk_bits(kx, kb);
                    store8(Ks + key * LD + dc * 8, kb);
#pragma unroll
                    for (int j = 0; j < 8; ++j) x3_store_vt(reinterpret_cast<float*>(vdst + ((j & 3) * 16 + (j >> 2)) * LDP), key, (float)frag_elem(vraw[i], j));
                } else {
                    store8(Ks + key * LD + dc * 8, kx);
#pragma unroll
                    for (int j = 0; j < 8; ++j) vdst[((j & 3) * 16 + (j >> 2)) * LDP] = frag_elem(vraw[i], j);   // V moves as stored: no conversion
                }
            } else {
                const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (key < nsub * 16) store8(Ks + key * LD + dc * 8, z);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (X3) x3_store_vt(reinterpret_cast<float*>(vdst + ((j & 3) * 16 + (j >> 2)) * LDP), key, 0.f);
                    else vdst[((j & 3) * 16 + (j >> 2)) * LDP] = to_act<T>(0.f);
                }
            }
        }
    }
    __syncthreads();

    const int np = a.npad ? a.npad[win] : 0;
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
        const int mt = wave + 4 * u;
        if (mt >= nmt) break;
        // Q fragment of this 16-row tile (RoPE + scale in f32); B operand: column = query r16
        Frag<T> qf[2];
        X3Frag qx[2];
        {
            const bool ok = mt * 16 + r16 < n;
            const int qrc = ok ? s_slots[mt * 16 + r16] : 0;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int d0 = s2 * 32 + g * 8;
                float x[8];
                frag_to_float(qraw[u][s2], x);
                rope8_lds(x, s_cos, s_sin, L, qrc, d0 >> 3);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = ok ? x[j] * a.scale : 0.f;
                if constexpr (X3) qx[s2] = x3_split(x);
                else qf[s2] = make_frag(x, T());
            }
        }
        // S^T = K Q^T over all keys; lane holds S[q = r16][key = t*16 + g*4 + r]
        f32x4 sc[MAXSUB];
        float mx = np > 0 ? 0.f : NEG_BIG;
#pragma unroll
        for (int t = 0; t < MAXSUB; ++t) {
            sc[t] = f32x4{NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG};
            if (t < nsub) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if constexpr (X3) x3_mma(acc, x3_from_chunk(read_frag(reinterpret_cast<const float*>(Ks) + (t * 16 + r16) * LD + s2 * 32 + g * 8)), qx[s2]);
                    else mma_step(acc, read_frag(Ks + (t * 16 + r16) * LD + s2 * 32 + g * 8), qf[s2]);
                }
                if (t * 16 + 16 > nkeys) {                        // only the last tile can straddle the end of the key list (wave-uniform)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = t * 16 + g * 4 + r < nkeys ? acc[r] : NEG_BIG;
                }
                sc[t] = acc;
                mx = fmaxf(mx, fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])));
            }
        }
        mx = g4_max(mx);
        // O = P V with P = exp(S - max) taken straight from the score registers (slot numbering of the header comment)
        f32x4 o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < (MAXSUB + 1) / 2; ++c) {
            if (c * 32 < NK32) {
                float pv[8];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = 2 * c + tt;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float e = 0.f;
                        if (t < MAXSUB && t < nsub) e = softmax_exp<T>(sc[t < MAXSUB ? t : 0][r] - mx);
                        pv[tt * 4 + r] = e;
                    }
                }
                Frag<T> pf;
                X3Frag px;
                if constexpr (X3) px = x3_split(pv);
                else pf = make_frag(pv, T());
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += pv[j];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const T* vrow = Vt + (d * 16 + r16) * LDP + c * 32 + g * 4;
                    if constexpr (X3) x3_mma(o[d], px, x3_from_halves(reinterpret_cast<const float*>(vrow), reinterpret_cast<const float*>(vrow) + 16));
                    else mma_step(o[d], pf, frag_from_halves(vrow, vrow + 16));
                }
            }
        }
        sum = g4_sum(sum);
        float padw = 0.f;
        if (np > 0) { padw = (float)np * softmax_exp<T>(-mx); sum += padw; }
        const float inv = 1.f / sum;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = mt * 16 + g * 4 + r;
            const float inv_q = __shfl(inv, g * 4 + r, 64), padw_q = __shfl(padw, g * 4 + r, 64);   // lane q holds query q's row state
            if (qi < n) {
                T* dst = reinterpret_cast<T*>(a.out) + (int64_t)s_rows[qi] * a.ldo + head * HD + r16 * 4;
                T o4[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {                   // V^T row d*16 + r16 holds head dim r16*4 + d (vt_row)
                    float v = o[d][r];
                    if (np > 0) v += padw_q * a.v_bias[head * HD + r16 * 4 + d];
                    o4[d] = to_act<T>(v * inv_q);
                }
                if constexpr (sizeof(T) == 4) {
                    if (a.out_planes) store4_planes(dst, o4);
                    else store4(dst, o4);
                } else {
                    store4(dst, o4);
                }
            }
        }
    }
}

template <typename T>
size_t attn_small_lds(int64_t stride, int L) {
    const int64_t nsub = (stride + 15) / 16, nk32 = (stride + 31) / 32 * 32, ldp = nk32 + 16 / (int)sizeof(T);
    return (size_t)(nsub * 16 * Pad<T>::ld + HD * ldp) * sizeof(T) + (size_t)((stride + 3) & ~3) * 8 + (size_t)L * 16 * 16;
}

template <typename T, int MAXSUB, bool X3 = false>
void launch_small(const AttnArgs& a, size_t lds, int64_t num_heads, int64_t nwin, hipStream_t s) {
    static Toc3dLdsAttr attr;                    // per function instantiation, per device
    attr.ensure(reinterpret_cast<const void*>(&attn_small_kernel<T, MAXSUB, X3>), 96 * 1024);
    toc3d_launch((attn_small_kernel<T, MAXSUB, X3>), dim3((unsigned)num_heads, (unsigned)(nwin + a.pf_rows)), dim3(256), lds, s, a);
}

template <typename T, bool X3 = false>
void launch_attn(const AttnArgs& a, int64_t max_count, int64_t num_heads, int64_t nwin, hipStream_t s) {
    // windows of up to 208 keys (virtual kept-pad keys included; every accelerated block of the shipped configs): the whole-window-resident
    // kernel, one workgroup per (window, head), instantiated for 144 / 176 / 208 keys (score registers per lane).  Measured r02: the dense
    // 16x16 windows (256 keys, 48 windows x 16 heads) run 49 us here against 45 us on the flash kernel below (more, smaller workgroups).
    if (a.stride <= 208) {
        const size_t lds = attn_small_lds<T>(a.stride, a.L);
        if (lds <= 96 * 1024) {
            if (a.stride <= 144) launch_small<T, 9, X3>(a, lds, num_heads, nwin, s);
            else if (a.stride <= 176) launch_small<T, 11, X3>(a, lds, num_heads, nwin, s);
            else launch_small<T, 13, X3>(a, lds, num_heads, nwin, s);
            return;
        }
    }
    // larger windows: flash-style kernel, 64-query workgroups (measured faster than or equal to 128-query ones on every window size of this
    // model, again in round 2 with P in registers: 256 keys 58.1 vs 55.9 us, 400 keys 60.9 vs 67.8 us, frames/s equal)
    const size_t lds = (size_t)(KT + HD) * Pad<T>::ld * sizeof(T) + (size_t)((a.stride + 3) & ~3) * 8 + (size_t)a.L * 16 * 4 * 4;
    dim3 grid((unsigned)((max_count + 63) / 64), (unsigned)num_heads, (unsigned)(nwin + a.pf_rows));
    toc3d_launch((attn_kernel<T, 1, X3>), grid, dim3(256), lds, s, a);
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

int toc3d_window_attention_pf(int dtype, const void* qkv, int64_t ldqkv, void* out, int64_t ldo, const int32_t* rows,
                              const int32_t* slots, const int32_t* count, const int32_t* count_k, const int32_t* npad,
                              const void* pad_qkv, int64_t stride, int64_t nwin, int64_t max_count, int64_t num_heads,
                              const float* rope_cos, const float* rope_sin, int64_t rope_side, const float* v_bias, float scale,
                              int64_t n_prefetch, const void* const* prefetch_ptrs, const int64_t* prefetch_bytes, int64_t prefetch_workgroups,
                              toc3d_stream_t stream) {
    TOC3D_REQUIRE(dtype == TOC3D_F32 || dtype == TOC3D_BF16 || dtype == TOC3D_F32X3 || dtype == TOC3D_F32X3P || dtype == TOC3D_F32X3WO, "toc3d_window_attention: bad dtype %d", dtype);
    // f32 q|k|v, f32 RoPE / softmax / accumulation in all of these: F32X3 / F32X3P = the two contractions as bf16 x 3 products; F32X3P / F32X3WO = the
    // output rows as (hi, lo) planes (the projection GEMM's A operand)
    const bool out_planes = dtype == TOC3D_F32X3P || dtype == TOC3D_F32X3WO, x3_products = dtype == TOC3D_F32X3 || dtype == TOC3D_F32X3P;
    if (dtype != TOC3D_BF16) dtype = TOC3D_F32;
    if (out_planes) {
        TOC3D_REQUIRE(((uintptr_t)out % 128) == 0 && ldo % 32 == 0, "toc3d_window_attention: rows of (hi, lo) planes start on 128-byte boundaries (out aligned, ldo a multiple of 32)");
    }
    TOC3D_REQUIRE(qkv && out && rows && slots && count && rope_cos && rope_sin, "toc3d_window_attention: null buffer");
    TOC3D_REQUIRE(!npad || v_bias, "toc3d_window_attention: npad given without v_bias");
    TOC3D_REQUIRE(!count_k || pad_qkv, "toc3d_window_attention: count_k given without pad_qkv");
    TOC3D_REQUIRE(num_heads > 0 && nwin >= 0 && max_count >= 0 && stride >= max_count, "toc3d_window_attention: bad dims");
    TOC3D_REQUIRE(rope_side > 0 && rope_side <= 256, "toc3d_window_attention: rope_side out of range");
    const int64_t C = num_heads * HD;
    TOC3D_REQUIRE(ldqkv >= 3 * C && ldo >= C, "toc3d_window_attention: leading dims too small for head_dim 64");
    const int esz = dtype == TOC3D_BF16 ? 2 : 4;
    TOC3D_REQUIRE((ldqkv * esz) % 16 == 0 && ((uintptr_t)qkv % 16) == 0, "toc3d_window_attention: qkv rows must be 16-byte aligned");
    TOC3D_REQUIRE(ldo % 4 == 0 && ((uintptr_t)out % 16) == 0, "toc3d_window_attention: out must be 16-byte aligned with ldo a multiple of 4");
    TOC3D_REQUIRE(num_heads <= 65535 && nwin <= 65535 - 128, "toc3d_window_attention: grid too large");
    if (nwin == 0 || max_count == 0) return TOC3D_OK;
    TOC3D_REQUIRE(n_prefetch >= 0 && n_prefetch <= 4 && (n_prefetch == 0 || (prefetch_ptrs && prefetch_bytes)), "toc3d_window_attention: at most 4 prefetch buffers (host arrays)");
    AttnArgs a{qkv, ldqkv, out, ldo, rows, slots, count, count_k, npad, pad_qkv, stride, (int)C, (int)rope_side, rope_cos, rope_sin, v_bias, scale,
               {nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}, (int)nwin, 0, out_planes ? 1 : 0};
    int64_t pf_total = 0;
    for (int64_t i = 0; i < n_prefetch; ++i) {
        TOC3D_REQUIRE(prefetch_bytes[i] >= 0 && ((uintptr_t)prefetch_ptrs[i] % 16) == 0, "toc3d_window_attention: prefetch buffers must be 16-byte aligned");
        a.pf_ptr[i] = (const f32x4*)prefetch_ptrs[i];
        a.pf_n16[i] = prefetch_bytes[i] / 16;
        pf_total += a.pf_n16[i];
    }
    if (pf_total > 0) {
        // prefetch workgroups come in rows of the grid's window dimension: at least num_heads workgroups per row
        const int64_t per_row = num_heads;
        int64_t rows_pf = ((prefetch_workgroups > 0 ? prefetch_workgroups : 256) + per_row - 1) / per_row;
        a.pf_rows = (int)(rows_pf < 1 ? 1 : (rows_pf > 128 ? 128 : rows_pf));
    }
    if (dtype == TOC3D_BF16) launch_attn<bf16_t>(a, max_count, num_heads, nwin, as_stream(stream));
    else if (x3_products) launch_attn<float, true>(a, max_count, num_heads, nwin, as_stream(stream));
    else launch_attn<float>(a, max_count, num_heads, nwin, as_stream(stream));
    TOC3D_LAUNCH_CHECK("toc3d_window_attention");
    return TOC3D_OK;
}

int toc3d_window_attention(int dtype, const void* qkv, int64_t ldqkv, void* out, int64_t ldo, const int32_t* rows,
                           const int32_t* slots, const int32_t* count, const int32_t* count_k, const int32_t* npad,
                           const void* pad_qkv, int64_t stride, int64_t nwin, int64_t max_count, int64_t num_heads,
                           const float* rope_cos, const float* rope_sin, int64_t rope_side, const float* v_bias, float scale,
                           toc3d_stream_t stream) {
    return toc3d_window_attention_pf(dtype, qkv, ldqkv, out, ldo, rows, slots, count, count_k, npad, pad_qkv, stride, nwin, max_count, num_heads,
                                     rope_cos, rope_sin, rope_side, v_bias, scale, 0, nullptr, nullptr, 0, stream);
}

int toc3d_window_map_dense(int64_t V, int64_t h, int64_t w, int64_t L, int32_t* rows, int32_t* slots, int32_t* count,
                           int32_t* npad, toc3d_stream_t stream) {
    TOC3D_REQUIRE(rows && slots && count && npad, "toc3d_window_map_dense: null buffer");
    TOC3D_REQUIRE(V > 0 && h > 0 && w > 0 && L > 0, "toc3d_window_map_dense: bad dims");
    const int nW = (int)(V * ((h + L - 1) / L) * ((w + L - 1) / L));
    toc3d_launch(window_map_dense_kernel, dim3(nW), dim3(256), 0, as_stream(stream), (int)V, (int)h, (int)w, (int)L, rows, slots, count, npad);
    TOC3D_LAUNCH_CHECK("toc3d_window_map_dense");
    return TOC3D_OK;
}

}  // extern "C"
