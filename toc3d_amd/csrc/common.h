// Shared device helpers for the gfx950 (CDNA4) kernels of the ToC3D backbone hot path.
// Wavefront = 64 lanes everywhere; MFMA fragments follow the 16x16 shapes:
//   bf16: v_mfma_f32_16x16x32_bf16   (A/B: 8 consecutive K elements per lane, lane = (row|col) + 16*kgroup)
//   f32 : v_mfma_f32_16x16x4_f32     (A/B: 1 element per lane, k = lane>>4), issued 8x per 32-wide K step
// Both element types use the same "8 consecutive K elements per lane" fragment so one tile loader /
// LDS layout serves the bf16 (throughput) and f32 (strict-parity) instantiations: the K order inside
// a step is permuted identically for A and B, which leaves the dot product unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define TOC3D_DEV __device__ __forceinline__

enum { TOC3D_F32 = 0, TOC3D_BF16 = 1, TOC3D_F32X3 = 2, TOC3D_F32X6 = 3, TOC3D_F32X3W = 4, TOC3D_F32X3P = 5, TOC3D_F32X3WO = 6, TOC3D_F32X3WA = 7 };       // F32X3: f32 buffers, GEMM products as three bf16 MFMAs (linear layers only)

// ---- element traits -------------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { f32x4 lo, hi; };

template <typename T> TOC3D_DEV T to_act(float x);
template <> TOC3D_DEV float to_act<float>(float x) { return x; }
template <> TOC3D_DEV bf16_t to_act<bf16_t>(float x) { return (bf16_t)x; }   // RNE, lowers to v_cvt_pk_bf16_f32

TOC3D_DEV float from_act(float x) { return x; }
TOC3D_DEV float from_act(bf16_t x) { return (float)x; }

// 8 consecutive elements <-> 8 floats (16-byte aligned for bf16, 32-byte for f32)
TOC3D_DEV void load8(const float* p, float (&o)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
TOC3D_DEV void load8(const bf16_t* p, float (&o)[8]) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)a[i];
}
TOC3D_DEV void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
TOC3D_DEV void store8(bf16_t* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
}
TOC3D_DEV Frag<float> make_frag(const float (&v)[8], float) {
    Frag<float> f; f.lo = f32x4{v[0], v[1], v[2], v[3]}; f.hi = f32x4{v[4], v[5], v[6], v[7]}; return f;
}
TOC3D_DEV Frag<bf16_t> make_frag(const float (&v)[8], bf16_t) {
    Frag<bf16_t> f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f.v[i] = (bf16_t)v[i];
    return f;
}
// read a fragment (8 consecutive elements) from LDS/global at a 16-byte aligned address
TOC3D_DEV Frag<bf16_t> read_frag(const bf16_t* p) { Frag<bf16_t> f; f.v = *reinterpret_cast<const bf16x8*>(p); return f; }
TOC3D_DEV Frag<float> read_frag(const float* p) {
    Frag<float> f; f.lo = *reinterpret_cast<const f32x4*>(p); f.hi = *reinterpret_cast<const f32x4*>(p + 4); return f;
}

TOC3D_DEV bf16_t gemm_frag_elem(const Frag<bf16_t>& f, int e) { return f.v[e]; }
TOC3D_DEV float gemm_frag_elem(const Frag<float>& f, int e) { return e < 4 ? f.lo[e] : f.hi[e - 4]; }

// one 32-wide K step of a 16x16 output tile
TOC3D_DEV void mma_step(f32x4& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
TOC3D_DEV void mma_step(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[s], b.lo[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[s], b.hi[s], acc, 0, 0, 0);
}

// ---- wave / block reductions ----------------------------------------------------------------------
// Cross-lane exchanges WITHOUT the LDS (round 5).  __shfl_xor lowers to ds_bpermute_b32 on gfx950: an LDS round trip (~100+ cycles of latency) per step, and the
// butterflies below sit on dependent chains -- the row kernels' LayerNorm (two 6-step reductions per row), every 32-key chunk of the attention, the statistics epilogues
// of the folded LayerNorms (32-64 exchanges per wavefront), the row table in front of the consuming GEMMs' K loops.  DPP moves (inside a 16-lane row) and the gfx950
// v_permlane16_swap / v_permlane32_swap (across rows) are plain VALU instructions: a few cycles each.  Every step pairs EXACTLY the lanes the shuffle form paired
// (l ^ 32, l ^ 16, l ^ 8, l ^ 4, l ^ 2, l ^ 1, in that order), so every sum keeps its association and its bits: the strict-parity path's near-tie top-k decisions at the
// 1600-wide inputs (tests/test_gpu_parity_bf16.py::test_vitl_1600_fp32_matches_reference) depend on them.
//   xor 1, xor 2: quad_perm;  xor 4: row_shl:4 into banks 0 / 2 and row_shr:4 into banks 1 / 3;  xor 8: row_ror:8;
//   xor 16: v_permlane16_swap(v, v) leaves v[l] and v[l ^ 16] in its two results;  xor 32: v_permlane32_swap likewise.
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_SHL4 = 0x104, DPP_ROW_SHR4 = 0x114, DPP_ROW_ROR8 = 0x128;
template <int CTRL> TOC3D_DEV int dpp_mov_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
TOC3D_DEV int lane_xor4_i(int v) {
    int t = __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHL4, 0xf, 0x5, false);     // lanes 0-3, 8-11 of every row take lane + 4
    return __builtin_amdgcn_update_dpp(t, v, DPP_ROW_SHR4, 0xf, 0xa, false);      // lanes 4-7, 12-15 take lane - 4
}
TOC3D_DEV float lane_xor1(float v) { return __builtin_bit_cast(float, dpp_mov_i<DPP_QUAD_XOR1>(__builtin_bit_cast(int, v))); }
TOC3D_DEV float lane_xor2(float v) { return __builtin_bit_cast(float, dpp_mov_i<DPP_QUAD_XOR2>(__builtin_bit_cast(int, v))); }
TOC3D_DEV float lane_xor4(float v) { return __builtin_bit_cast(float, lane_xor4_i(__builtin_bit_cast(int, v))); }
TOC3D_DEV float lane_xor8(float v) { return __builtin_bit_cast(float, dpp_mov_i<DPP_ROW_ROR8>(__builtin_bit_cast(int, v))); }
struct LanePair { float a, b; };
TOC3D_DEV LanePair swap16(float v) {                     // {v[l], v[l ^ 16]} in one order or the other (for commutative uses)
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return LanePair{__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1])};
}
TOC3D_DEV LanePair swap32(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return LanePair{__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1])};
}
// reduce across the 16 lanes that share lane>>4 (one MFMA C-row group): every lane of the row ends with the row's value
TOC3D_DEV float row16_sum(float v) {
    v += lane_xor8(v); v += lane_xor4(v); v += lane_xor2(v); v += lane_xor1(v);
    return v;
}
TOC3D_DEV float row16_max(float v) {
    v = fmaxf(v, lane_xor8(v)); v = fmaxf(v, lane_xor4(v)); v = fmaxf(v, lane_xor2(v)); v = fmaxf(v, lane_xor1(v));
    return v;
}
// the 4 lane groups g = lane >> 4 that share lane & 15 (one row of a transposed MFMA C tile)
TOC3D_DEV float g4_sum(float v) {
    const LanePair p = swap16(v);
    const LanePair q = swap32(p.a + p.b);
    return q.a + q.b;
}
TOC3D_DEV float g4_max(float v) {
    const LanePair p = swap16(v);
    const LanePair q = swap32(fmaxf(p.a, p.b));
    return fmaxf(q.a, q.b);
}
TOC3D_DEV float wave_sum(float v) {                      // l ^ 32, l ^ 16, then the row: the shuffle form's order
    const LanePair p = swap32(v);
    const LanePair q = swap16(p.a + p.b);
    return row16_sum(q.a + q.b);
}
TOC3D_DEV float wave_max(float v) {
    const LanePair p = swap32(v);
    const LanePair q = swap16(fmaxf(p.a, p.b));
    return row16_max(fmaxf(q.a, q.b));
}
// sum over the 4 lanes of a quad, f64 (the row table of the folded LayerNorms: four threads per row): + l ^ 1, then + l ^ 2, like the shuffle form
TOC3D_DEV double quad_sum(double v) {
    auto mov = [](double x, auto CT) {
        constexpr int C = decltype(CT)::value;
        const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
        const unsigned lo = (unsigned)dpp_mov_i<C>((int)(unsigned)u);
        const unsigned hi = (unsigned)dpp_mov_i<C>((int)(unsigned)(u >> 32));
        return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    };
    v += mov(v, std::integral_constant<int, DPP_QUAD_XOR1>());
    v += mov(v, std::integral_constant<int, DPP_QUAD_XOR2>());
    return v;
}

// 4 consecutive elements as one 8-byte (bf16) / 16-byte (f32) store
TOC3D_DEV void store4(bf16_t* p, const bf16_t (&v)[4]) {
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
    *reinterpret_cast<bf16x4_t*>(p) = bf16x4_t{v[0], v[1], v[2], v[3]};
}
TOC3D_DEV void store4(float* p, const float (&v)[4]) { *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]}; }

// An f32 buffer held as (hi, lo) bf16 PLANES (include/toc3d.h, TOC3D_DTYPE_F32X3P: the A operands of the bf16 x 3 GEMM, written by their producers): the size
// and leading dimension of the f32 buffer; element c of a row lives in the 128-byte group c / 32 -- hi = bf16(x) at byte 2 (c % 32), lo = bf16(x - hi) at byte
// 64 + 2 (c % 32).  f32p_t tags such a buffer in the row kernels' templates (pointer arithmetic as for float).  Rows start on 128-byte boundaries (host-checked),
// so group and position follow from the address the f32 element would have.
struct f32p_t { float v; };
TOC3D_DEV void store4_planes(void* f32_addr, const float (&v)[4]) {      // 4 consecutive elements, the first at a multiple of 4
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bf16_t h = (bf16_t)v[e];
        hi[e] = h;
        lo[e] = (bf16_t)(v[e] - (float)h);
    }
    const uintptr_t a = reinterpret_cast<uintptr_t>(f32_addr);
    char* g = reinterpret_cast<char*>(a & ~(uintptr_t)127) + ((a & 127) >> 1);
    *reinterpret_cast<bf16x4_t*>(g) = hi;
    *reinterpret_cast<bf16x4_t*>(g + 64) = lo;
}

// bijective XCD-aware remap of a 1-D grid (cdna_hip_programming.md T1): blocks that land on one XCD
// (bid % 8) get a contiguous chunk of work ids so neighbouring tiles share that XCD's L2.
// the same for the blocks [off, off + n) of a grid (other blocks of the launch do something else): block bid runs on XCD bid % 8; returns its work id in [0, n)
TOC3D_DEV int xcd_remap_off(int bid, int off, int n) {
    if (n < 8) return bid - off;
    const int xcd = bid & 7;
    int base = 0, j0x = 0;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const int j0 = ((y - off) % 8 + 8) % 8;          // first block of [0, n) (as j = bid - off) on XCD y
        const int cnt = n > j0 ? (n - j0 + 7) / 8 : 0;
        if (y < xcd) base += cnt;
        if (y == xcd) j0x = j0;
    }
    return base + (bid - off - j0x) / 8;
}
TOC3D_DEV int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    if (nwg < nx) return bid;
    int xcd = bid % nx, idx = bid / nx;
    int q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
