// Linear layers of the ToC3D / EVA-02 backbone on MFMA tiles (gfx950).
//
//   out = epilogue(A[M,K] . W[N,K]^T + bias)         A, W: K-contiguous ("B^T" input, nn.Linear layout)
//
// Reference ops served (all are nn.Linear / 1x1-conv shaped):
//   q/k/v projections  eva_vit.py:97-99, toc3d_eva_vit.py:495-497   (fused N = 3C, bias = q_bias|0|v_bias)
//   attn.proj          eva_vit.py:115          (+ residual add, eva_vit.py:262 / toc3d_eva_vit.py:379)
//   mlp.w1 | mlp.w2    eva_vit.py:45-47        (interleaved, epilogue silu(x1)*x2)
//   mlp.w3             eva_vit.py:49           (+ residual add, eva_vit.py:263 / toc3d_eva_vit.py:384)
//   patch_embed.proj   eva_utils.py:279-287    (im2col rows; epilogue + bias + abs-pos, toc3d_eva_vit.py:243-247)
//   scorer in_conv/out_conv  toc3d_utils.py:99-112  (epilogue exact GELU)
//
// Tile: 128x128 per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4x4 MFMA 16x16 tiles),
// K step = 128 bytes per row (64 bf16 / 32 f32).  Operands go HBM -> LDS with 16-byte
// global_load_lds (no VGPR round trip), double buffered; the LDS image is lane-linear, so the
// bank-conflict swizzle (16-byte chunk c of row r lives at position c ^ (r & 7)) is applied on the
// *source* address and undone on the ds_read side (cdna_hip_programming.md rule 21).
// Workgroup ids are remapped XCD-aware so the tiles of one A row-panel share an L2.
//
// This header holds the kernels and the per-epilogue launch table; gemm_epi_*.hip instantiate it for disjoint groups of epilogues (so the
// groups compile in parallel: one translation unit took 3.5 minutes), gemm.hip holds the C ABI, the packing and layout kernels.
#pragma once
#include "capi.h"
#include "common.h"

struct GemmArgs {
    const void* A; int64_t lda;
    const void* W; int64_t ldw;
    const float* bias;
    void* out; int64_t ldo;
    const float* res; int64_t ldr; int res_mod;
    const int32_t* res_index;                            // residual row of output row m = res_index[m] (< 0: the output row itself, read in place)
    float* rep_out; const int32_t* rep_index;
    int M, N, K, n_valid;
    int order;                                           // 0: XCD chunks of row-major tiles; 1: per-XCD row band, m fastest
    int vec;                                             // epilogue may use 4-element vector accesses (alignment checked on the host)
    int vec8;                                            // bf16 outputs: rows are 16-byte aligned, so two row tiles may leave as one 16-byte store per lane
    // LayerNorms folded across GEMM boundaries (include/toc3d.h, toc3d_linear_fused): statistics this launch leaves / consumes
    float* stats; int stats_cap;                         // written:  int32 header [4] + f32 [M, stats_cap, 2]
    const float* stats_in; int stats_in_cap;             // consumed: same layout, written by the launch that produced A
    int stats_in_slots;                                  // slots per row the producer wrote, when the host knows it (0: read the header -- one more dependent round trip)
    const float* c1; float ln_inv_n, ln_eps;             // consumed side: column sums of the gamma-scaled W, 1 / (normalised width), eps
    void* out_act; int64_t ld_act;                       // EPI_RESIDUAL_STATS: act-dtype copy of the f32 output rows (the next GEMM's A operand)
    // EPI_CONV3X3: A is an NHWC act tensor [V, conv_h, conv_w, lda]; the kernel gathers the 3x3 (pad 1) patches itself, K = 9 * lda in (ky, kx, c) order
    int conv_h, conv_w; const void* zeros;               // zeros: >= 128 bytes of zeros (the out-of-image taps)
    // EPI_QKV_ROPE: N = 3C, columns [q | k | v]; q and k are rotated by the row's RoPE position, q scaled (include/toc3d.h, toc3d_linear_qkv_rope)
    const int32_t* rope_rc;                              // [M] (table row of dims 0..31) << 16 | (table row of dims 32..63)
    const float* rope_tab;                               // compact axial tables [cos | sin][2][rope_L][16]
    int rope_L; float rope_scale;
    // bf16 x 3 on (hi, lo) PLANES (TOC3D_DTYPE_F32X3W / F32X3P, include/toc3d.h): an operand that already is in the planes layout is DMA'd as it lies and
    // not split in LDS; out_planes: the act-dtype outputs that a later GEMM multiplies (SwiGLU hidden units, out_act) are written as planes
    int a_planes, w_planes, out_planes;
    // deterministic split-K (toc3d_linear_fused_ws, variants >= 1000): `split` workgroups per output tile, each over its own range of K; f32 partial tiles
    // through sk_slabs, arrival tickets in sk_tickets (zero before the first launch; the last arriver re-arms its word)
    int split; float* sk_slabs; unsigned* sk_tickets;
    // bf16 x 3 / parity-GRADE (not strict) f32 instantiations: the SwiGLU epilogue's SiLU by the hardware exp2 / rcp (~1 ulp each: 2e-7 relative, against the 2^-16 of the
    // products) instead of expf + IEEE division -- ~35 VALU instructions per hidden unit, 16 units per lane, a tenth of a w1|w2 tile's time.  Exact f32 keeps the precise form.
    int fast_silu;
};

extern thread_local bool g_bad_variant;                // set by a launch_cfg whose tile variant cannot serve the requested epilogue (gemm.hip)

// one entry per group of epilogues (gemm_epi_*.hip); is_bf16 selects the element type
int toc3d_gemm_launch_plain(int is_bf16, int epi, int variant, const GemmArgs& a, hipStream_t s);      // EPI_BIAS, EPI_GELU, EPI_CONV3X3
int toc3d_gemm_launch_residual(int is_bf16, int epi, int variant, const GemmArgs& a, hipStream_t s);   // EPI_RESIDUAL, EPI_RESIDUAL_LN, EPI_RESIDUAL_STATS
int toc3d_gemm_launch_swiglu(int is_bf16, int epi, int variant, const GemmArgs& a, hipStream_t s);     // EPI_SWIGLU, EPI_SWIGLU_STATS, EPI_SWIGLU_STATS_LN
int toc3d_gemm_launch_rope(int is_bf16, int epi, int variant, const GemmArgs& a, hipStream_t s);       // EPI_QKV_ROPE (bf16)
int toc3d_gemm_launch_x3(int epi, int variant, const GemmArgs& a, hipStream_t s);                      // bf16 x 3 products on f32 operands: epilogues 0-9
int toc3d_gemm_launch_x6(int epi, int variant, const GemmArgs& a, hipStream_t s);                      // bf16 x 6 (three-way split): f32-grade products
int toc3d_gemm_launch_splitk(int dtype, int epi, int variant, const GemmArgs& a, hipStream_t s);       // residual epilogues with a.split > 1 (gemm_epi_splitk.hip); dtype: TOC3D_BF16 / TOC3D_F32 / TOC3D_F32X3
int toc3d_gemm_splitk_tile_dims(int variant);                                                           // BM << 16 | BN of a split-K tile variant (0: the variant has no split-K form)

// Development instrumentation (tools/ubench/gemm_timeline.hip builds its own copy of these kernels with -DTOC3D_GEMM_TRACE; the library
// never defines it): every workgroup leaves the 100 MHz real-time counter at entry, after its K loop and after its epilogue stores have
// been acknowledged, plus the hardware id of the CU it ran on.
#ifdef TOC3D_GEMM_TRACE
__device__ unsigned long long* toc3d_trace_buf;       // [grid][4]
#define TOC3D_TRACE(slot)                                                                                                   \
    do {                                                                                                                    \
        if (threadIdx.x == 0 && toc3d_trace_buf) toc3d_trace_buf[(size_t)blockIdx.x * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define TOC3D_TRACE_END()                                                                                                   \
    do {                                                                                                                    \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                    \
        TOC3D_TRACE(2);                                                                                                     \
        if (threadIdx.x == 0 && toc3d_trace_buf)                                                                            \
            toc3d_trace_buf[(size_t)blockIdx.x * 4 + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4); \
    } while (0)
#else
#define TOC3D_TRACE(slot) do {} while (0)
#define TOC3D_TRACE_END() do {} while (0)
#endif

namespace {

// RB = bytes of K per LDS row per stage: 128 (64 bf16 / 32 f32 per K-tile) or 64 (32 bf16, bf16 only)



constexpr bool epi_is_swiglu(int epi) { return epi == TOC3D_EPI_SWIGLU || epi == TOC3D_EPI_SWIGLU_STATS || epi == TOC3D_EPI_SWIGLU_STATS_LN; }
constexpr bool epi_is_residual(int epi) { return epi == TOC3D_EPI_RESIDUAL || epi == TOC3D_EPI_RESIDUAL_LN || epi == TOC3D_EPI_RESIDUAL_STATS || epi == TOC3D_EPI_CONV3X3; }
constexpr bool epi_ln_stats_in(int epi) { return epi == TOC3D_EPI_RESIDUAL_LN || epi == TOC3D_EPI_SWIGLU_STATS_LN; }      // LayerNorm of the A rows folded into the epilogue, its statistics left by the producing GEMM
constexpr bool epi_ln_in(int epi) { return epi_ln_stats_in(epi); }
constexpr bool epi_is_rope(int epi) { return epi == TOC3D_EPI_QKV_ROPE; }
constexpr bool epi_act_copy(int epi) { return epi == TOC3D_EPI_RESIDUAL_STATS; }      // the residual epilogue that also leaves the rows in the act dtype
constexpr bool epi_stats_out(int epi) { return epi == TOC3D_EPI_SWIGLU_STATS || epi == TOC3D_EPI_SWIGLU_STATS_LN || epi == TOC3D_EPI_RESIDUAL_STATS; }
// statistics groups per wave-tile row: one per 32 packed columns (SwiGLU) or per 16 output columns (residual)
constexpr int epi_stat_groups(int epi, int NT) { return epi_is_swiglu(epi) ? (NT / 2 > 0 ? NT / 2 : 1) : NT; }

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS swizzle: 16-byte chunk c of row r is stored at chunk position c ^ swz(r).  128-byte rows: swz = r & 7;
// 64-byte rows (4 rows per 256-byte bank row): swz = (-(r >> 2)) & 3; rows of 256 / 512 bytes (every row starts on
// the same bank): swz = r & 15.  Each makes every ds_read_b128 lane group of the MFMA fragment reads hit 16
// distinct 16-byte slots.
template <int RB> TOC3D_DEV int swz(int r) { return RB >= 256 ? (r & 15) : (RB == 128 ? (r & 7) : ((4 - ((r >> 2) & 3)) & 3)); }

// stage one R-row x RB-byte operand tile with 16-byte global_load_lds: R*RB/16 chunks over 256 threads.
template <typename T, int R, int RB, int NTHR, int AUX = 0>
TOC3D_DEV void stage_tile(const T* __restrict__ g, int64_t ld, int row0, int max_row, int k0, char* lds_tile, int wave, int lane) {
    constexpr int CPR = RB / 16;                        // chunks per row
    // Tiles whose chunk count is not a multiple of the workgroup size (96- / 160-row tiles on 512 threads): the wavefronts past the end of the last round
    // re-issue pieces of the tile's head (same bytes to the same LDS place), so that EVERY wave issues the same number of DMA instructions -- the ring
    // variants count them with a constant s_waitcnt vmcnt(N).
    static_assert((R * CPR) % 64 == 0, "whole DMA instructions");
    constexpr int ITER = (R * CPR + NTHR - 1) / NTHR;
#pragma unroll
    for (int t = 0; t < ITER; ++t) {
        int base = t * NTHR + wave * 64;                // wave-uniform
        if constexpr ((R * CPR) % NTHR != 0) base = base < R * CPR ? base : base - R * CPR;
        const int cidx = base + lane;
        const int r = cidx / CPR, p = cidx % CPR;
        int gr = row0 + r;
        gr = gr < max_row ? gr : max_row;
        const char* src = reinterpret_cast<const char*>(g + (int64_t)gr * ld + k0) + ((p ^ swz<RB>(r)) << 4);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_tile + base * 16), 16, 0, AUX);
    }
}

// fragment of row r (tile-local) for the 32-wide K step s, lane group g = lane >> 4
template <int RB>
TOC3D_DEV Frag<bf16_t> lds_frag(const char* tile, int r, int s, int g, bf16_t) {
    const int cc = s * 4 + g;
    Frag<bf16_t> f;
    f.v = *reinterpret_cast<const bf16x8*>(tile + r * RB + ((cc ^ swz<RB>(r)) << 4));
    return f;
}
template <int RB>
TOC3D_DEV Frag<float> lds_frag(const char* tile, int r, int s, int g, float) {
    static_assert(RB >= 128, "f32 tiles use rows of >= 128 bytes");
    Frag<float> f;
    f.lo = *reinterpret_cast<const f32x4*>(tile + r * RB + (((s * 8 + 2 * g) ^ swz<RB>(r)) << 4));
    f.hi = *reinterpret_cast<const f32x4*>(tile + r * RB + (((s * 8 + 2 * g + 1) ^ swz<RB>(r)) << 4));
    return f;
}

TOC3D_DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// SiLU of the SwiGLU epilogue.  The precise expf + IEEE division are ~35 VALU instructions per element, 16 elements per lane in the
// epilogue of the frame's largest GEMM (w1|w2); the bf16 path rounds the product to 8 bits anyway, so it takes the hardware exp2 / rcp
// (v_exp_f32, v_rcp_f32: ~1 ulp) -- the strict-parity f32 instantiation keeps the precise forms.
template <typename T> TOC3D_DEV float silu(float x);
template <> TOC3D_DEV float silu<float>(float x) { return x / (1.0f + expf(-x)); }
template <> TOC3D_DEV float silu<bf16_t>(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

template <int N> TOC3D_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Workgroup barrier of the K loop.  __builtin_amdgcn_s_barrier() alone does not order memory operations for the compiler: without
// the fences the scheduler may hoist the next K-tile's global_load_lds above the barrier (or sink this tile's ds_reads below it),
// and another wavefront then reads an operand piece that is being overwritten -- 8-row pieces of a tile came out wrong a few times
// per thousand launches, only with other kernels co-resident on the CU (tests/test_gpu_ops.py::test_linear_is_bit_stable_under_load).
TOC3D_DEV void tile_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
}

// Barrier that also publishes this wave's LDS stores (the raw barrier above does not wait for them) -- but NOT its global stores: a
// __syncthreads() here would hold every wave until its epilogue stores are acknowledged (s_waitcnt vmcnt(0)), microseconds per workgroup.
TOC3D_DEV void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tile_barrier();
}

// Epilogue stores.  TOC3D_WT_STORES (experiment, profiles/r02_write_through_stores.txt): write-through (sc1) stores, so the output does not sit
// dirty in the XCD's L2 until the end-of-kernel release writes it back in one burst.
#ifdef TOC3D_WT_STORES
TOC3D_DEV void epi_store4(bf16_t* p, const bf16_t (&v)[4]) {
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
    const bf16x4_t x = bf16x4_t{v[0], v[1], v[2], v[3]};
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
TOC3D_DEV void epi_store4(float* p, const float (&v)[4]) {
#if TOC3D_WT_STORES >= 2
    const f32x2 lo = f32x2{v[0], v[1]}, hi = f32x2{v[2], v[3]};
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, __builtin_bit_cast(unsigned long long, hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    store4(p, v);
#endif
}
#elif defined(TOC3D_NT_STORES)
// experiment (round 5, profiles/r05_nt_stores.txt): non-temporal epilogue stores (1: act-dtype outputs, 2: f32 outputs too)
TOC3D_DEV void epi_store4(bf16_t* p, const bf16_t (&v)[4]) {
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
    const bf16x4_t x = bf16x4_t{v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(__builtin_bit_cast(unsigned long long, x), reinterpret_cast<unsigned long long*>(p));
}
TOC3D_DEV void epi_store4(float* p, const float (&v)[4]) {
#if TOC3D_NT_STORES >= 2
    __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(p));
#else
    store4(p, v);
#endif
}
#else
TOC3D_DEV void epi_store4(bf16_t* p, const bf16_t (&v)[4]) { store4(p, v); }
TOC3D_DEV void epi_store4(float* p, const float (&v)[4]) { store4(p, v); }
#endif

// Wide bf16 epilogue stores (cdna_hip_programming.md T21, re-derived for the 16x16 C^T layout).  The epilogues of the single-round launches
// are store-ISSUE bound (all workgroups store at once; ~7 B/clk/CU whatever the byte count, profiles/r03_gemm_timeline_*.txt), and a bf16
// tile leaves as 8 bytes per lane.  Two row tiles i, i + 1 of the same columns are therefore merged into ONE 16-byte store per lane:
// v_permlane16_swap exchanges the odd 16-lane rows of `a` (tile i) with the even rows of `b` (tile i + 1); afterwards a lane of an even lane
// group g holds [own tile-i columns g*4.. | its right neighbour's], i.e. 8 consecutive columns of row i*16 + r16, and a lane of an odd group
// holds [left neighbour's | own] tile-(i+1) columns (g-1)*4.. of row (i+1)*16 + r16.  Same bytes, same addresses, half the instructions.
struct Pack4 { unsigned x, y; };
TOC3D_DEV Pack4 pack4(const bf16_t (&v)[4]) {
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
    const bf16x4_t q = bf16x4_t{v[0], v[1], v[2], v[3]};
    const unsigned long long u = __builtin_bit_cast(unsigned long long, q);
    return Pack4{(unsigned)u, (unsigned)(u >> 32)};
}
TOC3D_DEV Pack4 pack4(const float (&)[4]) { return Pack4{0u, 0u}; }     // f32 outputs already leave as 16 bytes per lane
// pa = this lane's 4 columns of row tile i, pb = of row tile i + 1 (both computed by every lane); dst_a / dst_b = the addresses this lane would
// have stored them to (8-byte stores); ok_a / ok_b = row inside the matrix
TOC3D_DEV void store_pair_wide(bf16_t* dst_a, bf16_t* dst_b, Pack4 pa, Pack4 pb, bool ok_a, bool ok_b, int g) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const auto rx = __builtin_amdgcn_permlane16_swap(pa.x, pb.x, false, false);
    const auto ry = __builtin_amdgcn_permlane16_swap(pa.y, pb.y, false, false);
    // even group: row tile i, columns start at its own; odd group: row tile i + 1, columns start 4 to the left (the left neighbour's)
    bf16_t* dst = (g & 1) ? dst_b - 4 : dst_a;
#ifdef TOC3D_NT_STORES
    if ((g & 1) ? ok_b : ok_a) __builtin_nontemporal_store(u32x4{rx[0], ry[0], rx[1], ry[1]}, reinterpret_cast<u32x4*>(dst));
#else
    if ((g & 1) ? ok_b : ok_a) *reinterpret_cast<u32x4*>(dst) = u32x4{rx[0], ry[0], rx[1], ry[1]};
#endif
}
TOC3D_DEV void store_pair_wide(float*, float*, Pack4, Pack4, bool, bool, int) {}

// (hi, lo) bf16 planes of a row of f32 values (the A / W operand layout of the bf16 x 3 GEMM once split: split_rows_x3 below): element c of the row lives in
// the 128-byte group c / 32 -- hi = bf16(x) at byte 2 (c % 32), lo = bf16(x - hi) at byte 64 + 2 (c % 32).  Same bytes per row as f32, same arithmetic as the
// in-LDS split, so a GEMM on planes returns the bits of the GEMM that splits.  col % 4 == 0.
TOC3D_DEV void store_planes4(float* row, int col, const float (&v)[4]) {
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bf16_t h = (bf16_t)v[e];
        hi[e] = h;
        lo[e] = (bf16_t)(v[e] - (float)h);
    }
    char* p = reinterpret_cast<char*>(row) + (col >> 5) * 128 + (col & 31) * 2;
    *reinterpret_cast<bf16x4_t*>(p) = hi;
    *reinterpret_cast<bf16x4_t*>(p + 64) = lo;
}
TOC3D_DEV void store_planes1(float* row, int col, float v) {
    const bf16_t h = (bf16_t)v;
    char* p = reinterpret_cast<char*>(row) + (col >> 5) * 128 + (col & 31) * 2;
    *reinterpret_cast<bf16_t*>(p) = h;
    *reinterpret_cast<bf16_t*>(p + 64) = (bf16_t)(v - (float)h);
}

// Two row tiles of a planes output at a time (round 6): the epilogues of the single-round launches are store-ISSUE bound (see store_pair_wide), and a planes tile
// left as FOUR 8-byte stores per lane and row-tile pair (hi and lo of each tile) where the f32 output it replaces took two 16-byte ones -- the rotating x3 q|k|v
// epilogue read 120 us against ~105 for the plain bias epilogue at M = 6000.  Here the hi planes of row tiles i, i + 1 leave as one 16-byte store per lane (the lane-group
// exchange of store_pair_wide on the bf16 hi values), the lo planes as another.  va / vb: this lane's 4 columns col.. of the two row tiles; col % 4 == 0.
TOC3D_DEV void store_planes_pair(float* row_a, float* row_b, int col, const float (&va)[4], const float (&vb)[4], bool ok_a, bool ok_b, int g) {
    bf16_t ha[4], la[4], hb[4], lb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ha[e] = (bf16_t)va[e]; la[e] = (bf16_t)(va[e] - (float)ha[e]);
        hb[e] = (bf16_t)vb[e]; lb[e] = (bf16_t)(vb[e] - (float)hb[e]);
    }
    const int off = (col >> 5) * 128 + (col & 31) * 2;
    bf16_t* pa = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(row_a) + off);
    bf16_t* pb = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(row_b) + off);
    store_pair_wide(pa, pb, pack4(ha), pack4(hb), ok_a, ok_b, g);
    store_pair_wide(pa + 32, pb + 32, pack4(la), pack4(lb), ok_a, ok_b, g);
}

// ---- epilogue of one wavefront's (MT*16) x (NT*16) accumulator block whose first row / column are row0 / col0.  The MFMA is issued
// with the operands swapped (W fragment as A, activation fragment as B), so a lane holds C[row = .. + r16][4 consecutive cols =
// .. + g*4 + 0..3]: 8-byte (bf16) / 16-byte (f32) vector accesses instead of 2- / 4-byte scattered ones.  a.vec (host-checked
// alignment / leading dims) enables the vector path. ----
// Statistics-writing epilogues: gs / gq [MT * G] (G = epi_stat_groups) receive, per row tile i and column group, this lane's share of
// (sum, sum of squares) over the ROUNDED act-dtype values it wrote (zero for rows / columns outside the matrix); the caller completes the sums.
// LayerNorm-consuming epilogues: lnrow = (mean, rstd) per tile row, prepared in LDS by the kernel.
template <typename T, int EPI, int MT, int NT>
TOC3D_DEV void gemm_epilogue(const GemmArgs& a, f32x4 (&acc)[MT][NT], int row0, int col0, int r16, int g, float* gs = nullptr,
                             float* gq = nullptr, const f32x2* lnrow = nullptr, const int* rope_rcs = nullptr, const float* rope_tab = nullptr) {
    constexpr int G = epi_stat_groups(EPI, NT);
    if (epi_is_swiglu(EPI)) {
        // packed columns: per 32-column group, cols 0-15 = w1 units, cols 16-31 = w2 of the same units
        T* out = reinterpret_cast<T*>(a.out);
        // hidden units (i, jp) of this lane -> hs; returns whether the lane's row / columns lie inside the matrix
        auto units = [&](int i, int jp, T (&hs)[4], float& ssum, float& sq) -> bool {
            const int row = row0 + i * 16 + r16;
            float mu = 0.f, rs = 1.f;
            if (epi_ln_in(EPI)) { const f32x2 v = lnrow[i * 16 + r16]; mu = v[0]; rs = v[1]; }
            const int pc = col0 + jp * 32 + g * 4;        // packed col of the w1 half, first of 4
            const int unit0 = (pc >> 5) * 16 + g * 4;
            ssum = 0.f; sq = 0.f;
            const bool in = pc < a.N && row < a.M;
#pragma unroll
            for (int r = 0; r < 4; ++r) hs[r] = to_act<T>(0.f);
            if (in) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x1, x2;
                    if (epi_ln_in(EPI)) {               // bias = c2 (beta . W + b), c1 = column sums of the gamma-scaled packed weights
                        x1 = rs * (acc[i][2 * jp][r] - mu * a.c1[pc + r]) + a.bias[pc + r];
                        x2 = rs * (acc[i][2 * jp + 1][r] - mu * a.c1[pc + 16 + r]) + a.bias[pc + 16 + r];
                    } else {
                        x1 = acc[i][2 * jp][r] + a.bias[pc + r];
                        x2 = acc[i][2 * jp + 1][r] + a.bias[pc + 16 + r];
                    }
                    float sv;
                    if constexpr (sizeof(T) == 4) sv = a.fast_silu ? silu<bf16_t>(x1) : silu<T>(x1);      // (wave-uniform)
                    else sv = silu<T>(x1);
                    hs[r] = to_act<T>(unit0 + r < a.n_valid ? sv * x2 : 0.f);
                    if (epi_stats_out(EPI)) {
                        const float hv = from_act(hs[r]);       // what the next GEMM multiplies: the rounded value
                        ssum += hv;
                        sq = __builtin_fmaf(hv, hv, sq);
                    }
                }
            }
            return in;
        };
        if constexpr (sizeof(T) == 2 && MT % 2 == 0) {
            if (a.vec8) {                                // pairs of row tiles leave as one 16-byte store per lane
#pragma unroll
                for (int i = 0; i < MT; i += 2)
#pragma unroll
                    for (int jp = 0; jp < NT / 2; ++jp) {
                        T ha[4], hb[4];
                        float s1, q1, s2, q2;
                        const bool ia = units(i, jp, ha, s1, q1), ib = units(i + 1, jp, hb, s2, q2);
                        if (epi_stats_out(EPI)) { gs[i * G + jp] = s1; gq[i * G + jp] = q1; gs[(i + 1) * G + jp] = s2; gq[(i + 1) * G + jp] = q2; }
                        const int unit0 = ((col0 + jp * 32 + g * 4) >> 5) * 16 + g * 4;
                        T* da = out + (int64_t)(row0 + i * 16 + r16) * a.ldo + unit0;
                        if (col0 + jp * 32 + 32 <= a.N) {                        // the whole 32-column group lies inside (wave-uniform): wide stores
                            // a lane's own in-flag stands for its neighbour's too: both share the row, and the columns are all inside
                            store_pair_wide(da, da + 16 * a.ldo, pack4(ha), pack4(hb), ia, ib, g);
                        } else {
                            if (ia) epi_store4(da, ha);
                            if (ib) epi_store4(da + 16 * a.ldo, hb);
                        }
                    }
                return;
            }
        }
        if constexpr (sizeof(T) == 4 && MT % 2 == 0) {
            if (a.out_planes && a.vec) {                 // hidden units as (hi, lo) planes: pairs of row tiles leave as 16-byte stores (store_planes_pair)
#pragma unroll
                for (int i = 0; i < MT; i += 2)
#pragma unroll
                    for (int jp = 0; jp < NT / 2; ++jp) {
                        T ha[4], hb[4];
                        float s1, q1, s2, q2;
                        const bool ia = units(i, jp, ha, s1, q1), ib = units(i + 1, jp, hb, s2, q2);
                        if (epi_stats_out(EPI)) { gs[i * G + jp] = s1; gq[i * G + jp] = q1; gs[(i + 1) * G + jp] = s2; gq[(i + 1) * G + jp] = q2; }
                        const int unit0 = ((col0 + jp * 32 + g * 4) >> 5) * 16 + g * 4;
                        float* ra = reinterpret_cast<float*>(out) + (int64_t)(row0 + i * 16 + r16) * a.ldo;
                        if (col0 + jp * 32 + 32 <= a.N) {                        // the whole 32-column group lies inside (wave-uniform)
                            store_planes_pair(ra, ra + 16 * a.ldo, unit0, ha, hb, ia, ib, g);
                        } else {
                            if (ia) store_planes4(ra, unit0, ha);
                            if (ib) store_planes4(ra + 16 * a.ldo, unit0, hb);
                        }
                    }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int jp = 0; jp < NT / 2; ++jp) {
                T hs[4];
                float ssum, sq;
                const bool in = units(i, jp, hs, ssum, sq);
                if (in) {
                    const int unit0 = ((col0 + jp * 32 + g * 4) >> 5) * 16 + g * 4;
                    T* dst = out + (int64_t)(row0 + i * 16 + r16) * a.ldo + unit0;
                    bool planes = false;
                    if constexpr (sizeof(T) == 4) {
                        if (a.out_planes) { store_planes4(reinterpret_cast<float*>(dst - unit0), unit0, hs); planes = true; }
                    }
                    if (planes) {}
                    else if (a.vec) epi_store4(dst, hs);
                    else { dst[0] = hs[0]; dst[1] = hs[1]; dst[2] = hs[2]; dst[3] = hs[3]; }
                }
                if (epi_stats_out(EPI)) { gs[i * G + jp] = ssum; gq[i * G + jp] = sq; }
            }
        }
        return;
    }
    float bcol[NT][4];
    constexpr bool LN_IN = epi_ln_in(EPI);               // (non-SwiGLU epilogues from here on)
    float ccol[LN_IN ? NT : 1][4];                       // c1: column sums of the gamma-scaled weights
    if (EPI == TOC3D_EPI_RESIDUAL_STATS) {
#pragma unroll
        for (int q = 0; q < MT * G; ++q) { gs[q] = 0.f; gq[q] = 0.f; }
    }
    int nok[NT];                                         // valid columns among the lane's 4 (0..4)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = col0 + j * 16 + g * 4;
        nok[j] = a.N - col < 0 ? 0 : (a.N - col > 4 ? 4 : a.N - col);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bcol[j][r] = (a.bias && r < nok[j]) ? a.bias[col + r] : 0.f;
            if (LN_IN) ccol[j][r] = r < nok[j] ? a.c1[col + r] : 0.f;
        }
    }
    // act-dtype outputs (bias / GELU / rotated q|k|v): the 4 values of row tile i, column tile j
    auto act4 = [&](int i, int j, T (&o4)[4]) {
        const int col = col0 + j * 16 + g * 4;
        if constexpr (epi_is_rope(EPI)) {
            // RoPE of the q and k columns on the f32 accumulators (eva_utils.py:378-379: out[2t] = x[2t] cos - x[2t+1] sin,
            // out[2t+1] = x[2t+1] cos + x[2t] sin, the pair sharing one frequency), q scaled afterwards (eva_vit.py:104-109): the
            // attention kernel then stages K and V by DMA with no arithmetic at all.  A lane's 4 columns are two whole pairs of one
            // head; a 16-column MFMA tile lies inside one of q / k / v (C is a multiple of 64), so the branch is wave-uniform.
            const int Cq = a.N / 3;
            const int rope_rc = rope_rcs[i];             // loaded before the K loop (no dependent global round trip here)
            float x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = acc[i][j][r] + bcol[j][r];
            if (col < 2 * Cq) {
                const int d0 = col & 63, part = d0 >> 5;
                const int coord = part ? (rope_rc & 0xffff) : (rope_rc >> 16);
                const int off = (part * a.rope_L + coord) * 16 + ((d0 & 31) >> 1);       // tables in LDS: [cos | sin], each [2][L][16]
                const f32x2 c2 = *reinterpret_cast<const f32x2*>(rope_tab + off), s2 = *reinterpret_cast<const f32x2*>(rope_tab + 2 * a.rope_L * 16 + off);
                const float y0 = x[0] * c2[0] - x[1] * s2[0], y1 = x[1] * c2[0] + x[0] * s2[0];
                const float y2 = x[2] * c2[1] - x[3] * s2[1], y3 = x[3] * c2[1] + x[2] * s2[1];
                const float sc = col < Cq ? a.rope_scale : 1.0f;
                x[0] = y0 * sc; x[1] = y1 * sc; x[2] = y2 * sc; x[3] = y3 * sc;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = to_act<T>(x[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float raw = acc[i][j][r] + bcol[j][r];
                o4[r] = to_act<T>(EPI == TOC3D_EPI_GELU ? gelu_erf(raw) : raw);
            }
        }
    };
    auto act_row = [&](int i, int row) {                 // one row tile, 8-byte (bf16) / 16-byte (f32) stores
        T* orow = reinterpret_cast<T*>(a.out) + (int64_t)row * a.ldo;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (nok[j] == 0) continue;
            const int col = col0 + j * 16 + g * 4;
            T o4[4];
            act4(i, j, o4);
            if constexpr (sizeof(T) == 4 && epi_is_rope(EPI)) {
                // bf16 x 3 on planes: the rotated q | k | v rows leave as (hi, lo) planes -- what toc3d_window_attention_rot stages by DMA (N = 3C, C % 64 == 0: whole groups of 4)
                if (a.out_planes) { store_planes4(reinterpret_cast<float*>(orow), col, o4); continue; }
            }
            if (a.vec && nok[j] == 4) epi_store4(orow + col, o4);
            else for (int r = 0; r < nok[j]; ++r) orow[col + r] = o4[r];
        }
    };
    if constexpr (sizeof(T) == 4 && epi_is_rope(EPI) && MT % 2 == 0) {
        if (a.out_planes) {                              // rotated q | k | v rows as planes, pairs of row tiles as 16-byte stores (N = 3C, C % 64 == 0: whole column tiles)
#pragma unroll
            for (int i = 0; i < MT; i += 2) {
                const int ra = row0 + i * 16 + r16, rb = ra + 16;
                float* oa = reinterpret_cast<float*>(a.out) + (int64_t)ra * a.ldo;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (col0 + j * 16 + 16 > a.N) continue;
                    T va[4], vb[4];
                    act4(i, j, va);
                    act4(i + 1, j, vb);
                    store_planes_pair(oa, oa + 16 * a.ldo, col0 + j * 16 + g * 4, va, vb, ra < a.M, rb < a.M, g);
                }
            }
            return;
        }
    }
    if constexpr (!epi_is_residual(EPI) && sizeof(T) == 2 && MT % 2 == 0) {
        if (a.vec8) {                                    // bf16 outputs: pairs of row tiles leave as one 16-byte store per lane (store_pair_wide)
#pragma unroll
            for (int i = 0; i < MT; i += 2) {
                const int ra = row0 + i * 16 + r16, rb = ra + 16;
                T* oa = reinterpret_cast<T*>(a.out) + (int64_t)ra * a.ldo;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int col = col0 + j * 16 + g * 4;
                    if (col0 + j * 16 + 16 <= a.N) {     // the whole 16-column tile lies inside (wave-uniform)
                        T va[4], vb[4];
                        act4(i, j, va);
                        act4(i + 1, j, vb);
                        store_pair_wide(oa + col, oa + 16 * a.ldo + col, pack4(va), pack4(vb), ra < a.M, rb < a.M, g);
                    } else if (nok[j] > 0) {
                        T o4[4];
                        if (ra < a.M) { act4(i, j, o4); for (int r = 0; r < nok[j]; ++r) oa[col + r] = o4[r]; }
                        if (rb < a.M) { act4(i + 1, j, o4); for (int r = 0; r < nok[j]; ++r) oa[16 * a.ldo + col + r] = o4[r]; }
                    }
                }
            }
            return;
        }
    }
    // The act-dtype copy of the residual epilogues as (hi, lo) planes: pairs of row tiles leave as 16-byte stores (store_planes_pair) -- the even tile's values are held until
    // the odd tile of the same columns is formed (rows ascend with i: a valid odd tile implies a valid even one; an even tile whose partner lies past M leaves alone)
    constexpr bool PAIR_OK = sizeof(T) == 4 && epi_act_copy(EPI) && epi_is_residual(EPI) && MT % 2 == 0;
    float keep[PAIR_OK ? NT : 1][4];
    float* keep_row = nullptr;
    bool pair = PAIR_OK && a.out_planes;
#pragma unroll
    for (int j = 0; j < NT; ++j) pair = pair && (nok[j] == 0 || nok[j] == 4);
    (void)keep; (void)keep_row;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = row0 + i * 16 + r16;
        if (row >= a.M) {
            if constexpr (PAIR_OK) {
                if (pair && (i & 1) && keep_row) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        if (nok[j] == 4) store_planes4(keep_row, col0 + j * 16 + g * 4, keep[j]);
                }
            }
            continue;
        }
        float mu = 0.f, rs = 1.f;                        // EPI_RESIDUAL_LN: (mean, rstd) of this A row, prepared in LDS by the kernel
        if (LN_IN && epi_is_residual(EPI)) { const f32x2 v = lnrow[i * 16 + r16]; mu = v[0]; rs = v[1]; }
        (void)mu; (void)rs;
        if (epi_is_residual(EPI)) {
            // the modular residual row and the representative-row test cost an integer division / a load each: once per row
            const int rr = a.res_mod > 0 ? row % a.res_mod : row;
            const float* resrow = a.res ? a.res + (int64_t)rr * a.ldr : nullptr;
            float* orow = reinterpret_cast<float*>(a.out) + (int64_t)row * a.ldo;
            if (a.res_index) {                           // compact rows whose residual still sits in the token-major stream (no f32 copy was made)
                const int ti = a.res_index[row];
                resrow = ti >= 0 ? a.res + (int64_t)ti * a.ldr : orow;
            }
            float* reprow = nullptr;
            if (a.rep_index) { const int ri = a.rep_index[row]; if (ri >= 0) reprow = a.rep_out + (int64_t)ri * a.N; }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (nok[j] == 0) continue;
                const int col = col0 + j * 16 + g * 4;
                float raw[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (LN_IN) raw[r] = rs * (acc[i][j][r] - mu * ccol[j][r]) + bcol[j][r];
                    else raw[r] = acc[i][j][r] + bcol[j][r];
                }
                float sum4[4] = {0.f, 0.f, 0.f, 0.f};
                if (a.vec && nok[j] == 4) {
                    f32x4 rv = resrow ? *reinterpret_cast<const f32x4*>(resrow + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum4[r] = rv[r] + raw[r];
                    epi_store4(orow + col, sum4);
                    if (reprow) *reinterpret_cast<f32x4*>(reprow + col) = f32x4{raw[0], raw[1], raw[2], raw[3]};
                } else {
                    for (int r = 0; r < nok[j]; ++r) {
                        sum4[r] = (resrow ? resrow[col + r] : 0.f) + raw[r];
                        orow[col + r] = sum4[r];
                        if (reprow) reprow[col + r] = raw[r];
                    }
                }
                if (epi_act_copy(EPI)) {
                    // the updated residual-stream row also leaves in the act dtype (the next GEMM's A operand, its LayerNorm folded into that
                    // GEMM), RESIDUAL_STATS: together with the sums of those rounded values
                    T* arow = reinterpret_cast<T*>(a.out_act) + (int64_t)row * a.ld_act + col;
                    T o4[4];
                    float ssum = 0.f, sq = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o4[r] = to_act<T>(r < nok[j] ? sum4[r] : 0.f);
                        const float hv = from_act(o4[r]);
                        ssum += hv;
                        sq = __builtin_fmaf(hv, hv, sq);
                    }
                    bool planes = false;
                    if constexpr (sizeof(T) == 4) {
                        if (a.out_planes) {
                            bool done = false;
                            if constexpr (PAIR_OK) {
                                if (pair) {
                                    if ((i & 1) == 0) {
#pragma unroll
                                        for (int r = 0; r < 4; ++r) keep[j][r] = o4[r];
                                        keep_row = reinterpret_cast<float*>(arow - col);
                                    } else {
                                        store_planes_pair(keep_row, reinterpret_cast<float*>(arow - col), col, keep[j], o4, true, true, g);
                                    }
                                    done = true;
                                }
                            }
                            if (done) {}
                            else if (nok[j] == 4) store_planes4(reinterpret_cast<float*>(arow - col), col, o4);
                            else for (int r = 0; r < nok[j]; ++r) store_planes1(reinterpret_cast<float*>(arow - col), col + r, o4[r]);
                            planes = true;
                        }
                    }
                    if (planes) {}
                    else if (nok[j] == 4) epi_store4(arow, o4);
                    else for (int r = 0; r < nok[j]; ++r) arow[r] = o4[r];
                    if (EPI == TOC3D_EPI_RESIDUAL_STATS) {
                        gs[i * G + j] = ssum;
                        gq[i * G + j] = sq;
                    }
                }
            }
        } else {
            act_row(i, row);
        }
    }
}

// Multi-stage pipeline: the LDS ring holds STAGES K-tiles; tile t+STAGES-1 is requested while tile t is
// multiplied, so a K step no longer exposes an HBM/L2 round trip.  The in-flight global_load_lds are
// tracked with a *counted* s_waitcnt vmcnt(N) and a raw s_barrier (a __syncthreads() would drain them to
// vmcnt(0), cdna_hip_programming.md "Pipelining across barriers").  One barrier per K-tile:
//   wait(tile t landed) -> s_barrier -> request tile t+STAGES-1 into the slot tile t-1 just left -> MFMAs on tile t
// X3 (T = float only): "bf16 x 3" products on f32 operands -- the parity-grade path at bf16 MFMA speed (include/toc3d.h, TOC3D_DTYPE_F32X3).
// Every f32 operand fragment is split in registers into hi = bf16(x) and lo = bf16(x - hi) (x - hi is exact; |x - hi - lo| <= 2^-18 |x|) and
// a . w is accumulated in f32 as hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16: 3 bf16 MFMAs (48 cycles) instead of 8 exact-f32 ones
// (256 cycles) per 16x16x32 step, relative error of a product <= ~2^-16 (the dropped lo.lo term and the roundings of lo).  Memory layout,
// loaders, LDS images and epilogues are those of the f32 instantiation.
TOC3D_DEV void split_bf16x3(const Frag<float>& f, bf16x8& hi, bf16x8& lo) {
    const float x[8] = {f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bf16_t h = (bf16_t)x[e];
        hi[e] = h;
        lo[e] = (bf16_t)(x[e] - (float)h);
    }
}
// X = 6: three-way split x = hi + mid + lo (3 x 8 mantissa bits = all 24 of an f32; both subtractions are exact), products
// hi.hi + (hi.mid + mid.hi) + (mid.mid + hi.lo + lo.hi): the dropped terms are <= 2^-26 of the product -- f32-grade results from 6 bf16
// MFMAs (96 cycles) instead of 8 f32 ones (256).
TOC3D_DEV void split_bf16x6(const Frag<float>& f, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    const float x[8] = {f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bf16_t h = (bf16_t)x[e];
        const float r1 = x[e] - (float)h;
        const bf16_t m = (bf16_t)r1;
        hi[e] = h;
        mid[e] = m;
        lo[e] = (bf16_t)(r1 - (float)m);
    }
}

// (mean, rstd) per tile row from the partial sums the producing GEMM left (include/toc3d.h), into the row table `lnrow` [BM] in LDS.  Four threads per row;
// thread part p sums slots p, p + 4, ... in sequence, the parts meet in two butterfly steps; f64.  The order is fixed: the bits do not depend on the tile variant.
template <int EPI, int BM, int NTHR>
TOC3D_DEV void ln_rows_prepare_fn(const GemmArgs& a, const int m0, f32x2* lnrow, const int tid) {
        const int nslots = a.stats_in_slots > 0 ? a.stats_in_slots : *reinterpret_cast<const int*>(a.stats_in);
        const f32x2* base = reinterpret_cast<const f32x2*>(a.stats_in + 4);
        for (int w = tid; w < BM * 4; w += NTHR) {
            const int r = w >> 2, part = w & 3;
            int row = m0 + r;
            row = row < a.M ? row : a.M - 1;
            const f32x2* sp = base + (int64_t)row * a.stats_in_cap;
            double s1 = 0.0, s2 = 0.0;
            // every slot of the thread requested before the first is added: the plain loop (load, wait, add per slot) was nslots / 4 = 4-11 DEPENDENT L2 round
            // trips in front of every workgroup's K loop (round 4: +7.5 us on the w1|w2 launch at M = 6000).  Unconditional loads (clamped index), the validity
            // test on the value: a per-element "load or not" makes hipcc branch and wait per element (cdna_hip_programming.md, traps (c)).  Same order of additions.
            // slots 0 .. 15 in front of w1|w2 (residual-stream statistics: C / 64 = 16), 0 .. 47 in front of w3 (hidden units: 43); the register-capped tiles
            // (OCC > 1: 64 / 80 / 128 registers for 8 / 6 / 4 waves per SIMD) run this at the very top of the tile, before the accumulators exist (PREP_EARLY)
            constexpr int PRE = epi_is_swiglu(EPI) ? 4 : 12;
            f32x2 pv[PRE > 0 ? PRE : 1];
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                const int sl = part + 4 * j;
                pv[j] = sp[sl < nslots ? sl : 0];           // (slot 0 always exists: stats_in_cap >= 1; `part` itself may lie past the row when it has < 4 slots)
            }
#pragma unroll
            for (int j = 0; j < PRE; ++j)
                if (part + 4 * j < nslots) { s1 += (double)pv[j][0]; s2 += (double)pv[j][1]; }
            for (int sl = part + 4 * PRE; sl < nslots; sl += 4) {
                const f32x2 v = sp[sl];
                s1 += (double)v[0];
                s2 += (double)v[1];
            }
            s1 = quad_sum(s1);                           // (p0 + p1) + (p2 + p3) in every lane of the quad: DPP moves, no LDS round trip in front of the K loop
            s2 = quad_sum(s2);
            const double mean = s1 * (double)a.ln_inv_n;
            double var = s2 * (double)a.ln_inv_n - mean * mean;      // biased variance (F.layer_norm)
            var = var > 0.0 ? var : 0.0;
            if (part == 0) lnrow[r] = f32x2{(float)mean, 1.0f / sqrtf((float)var + a.ln_eps)};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // table rows written before this wave reaches the K loop's first barrier
}

// The part of a tile behind its K loop: the fused epilogue of the workgroup's accumulators (+ the statistics hand-off of the folded LayerNorms).  Shared by
// gemm_tile and the phased big-tile kernel.  PRE_BARRIER: the K loop did not end on a workgroup barrier (the statistics overlay the operand stages at `smem`).
template <typename T, int EPI, int BM, int BN, int WM, int WN, bool PRE_BARRIER>
TOC3D_DEV void tile_finish(const GemmArgs& a, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], const int m0, const int n0, char* smem, const f32x2* lnrow,
                           const int* rope_rcs, const float* rope_tab) {
    constexpr int NTHR = 64 * WM * WN, TM = BM / WM, TN = BN / WN, MT = TM / 16, NT = TN / 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, g = lane >> 4;
    if constexpr (epi_stats_out(EPI)) {
        // Row statistics of the act-dtype values this launch wrote, for the LayerNorm folded into the next GEMM (include/toc3d.h).  A slot is
        // 128 packed columns (SwiGLU: 64 hidden units) or 64 output columns (residual), i.e. always four column groups of the epilogue, and is
        // built in ONE fixed tree for every tile variant: lane -> its 4 values in order; group = butterfly over the 4 lane groups;
        // slot = (g0 + g1) + (g2 + g3), combined through LDS whatever wave computed the groups.
        constexpr int G = epi_stat_groups(EPI, NT);      // groups per wave-tile row
        constexpr int GW = epi_is_swiglu(EPI) ? 32 : 16; // columns per group
        constexpr int SLOT = 4 * GW, GPT = BN / GW;      // columns per slot, groups per tile row
        static_assert(BN % SLOT == 0 && (epi_is_swiglu(EPI) ? NT % 2 == 0 : true), "statistics need N-tiles of whole slots");
        float gs[MT * G], gq[MT * G];
        gemm_epilogue<T, EPI, MT, NT>(a, acc, m0 + wm * TM, n0 + wn * TN, r16, g, gs, gq, lnrow + wm * TM);
        // `red` overlays the operand stages: every wave must be done reading them.  The single-buffer loop ends on that barrier already (behind its last
        // multiply); the rings end on a multiply.  (The row table lives behind the stages and is not touched.)
        if constexpr (PRE_BARRIER) tile_barrier();
        f32x2* red = reinterpret_cast<f32x2*>(smem);     // [GPT][BM]
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jp = 0; jp < G; ++jp) {
                const float s1 = g4_sum(gs[i * G + jp]), s2 = g4_sum(gq[i * G + jp]);
                if (g == 0) red[(wn * G + jp) * BM + wm * TM + i * 16 + r16] = f32x2{s1, s2};
            }
        lds_barrier();
        f32x2* data = reinterpret_cast<f32x2*>(a.stats + 4);
        const int nslots = (a.N + SLOT - 1) / SLOT;      // (a 256-wide tile at the right edge covers slots that do not exist: they would land in the next row's slots)
        for (int w = tid; w < BM * (GPT / 4); w += NTHR) {
            const int r = w % BM, sl = w / BM;
            const int row = m0 + r;
            if (row >= a.M || n0 / SLOT + sl >= nslots) continue;
            const f32x2 v0 = red[(4 * sl) * BM + r], v1 = red[(4 * sl + 1) * BM + r], v2 = red[(4 * sl + 2) * BM + r], v3 = red[(4 * sl + 3) * BM + r];
            data[(int64_t)row * a.stats_cap + n0 / SLOT + sl] = f32x2{(v0[0] + v1[0]) + (v2[0] + v3[0]), (v0[1] + v1[1]) + (v2[1] + v3[1])};
        }
        if (m0 == 0 && n0 == 0 && tid == 0) *reinterpret_cast<int*>(a.stats) = (a.N + SLOT - 1) / SLOT;
    } else if constexpr (epi_is_rope(EPI)) {
        gemm_epilogue<T, EPI, MT, NT>(a, acc, m0 + wm * TM, n0 + wn * TN, r16, g, nullptr, nullptr, lnrow + wm * TM, rope_rcs, rope_tab);
    } else if constexpr (epi_ln_in(EPI)) {
        gemm_epilogue<T, EPI, MT, NT>(a, acc, m0 + wm * TM, n0 + wn * TN, r16, g, nullptr, nullptr, lnrow + wm * TM);
    } else {
        gemm_epilogue<T, EPI, MT, NT>(a, acc, m0 + wm * TM, n0 + wn * TN, r16, g);
    }
}

// ---- deterministic split-K: the exchange of the partial accumulators ---------------------------------------------------------------------
// Every slice stores its f32 partial tile WRITE-THROUGH (sc1: the data leaves the XCD's L2 with the store, so no release fence -- cdna_hip_programming.md
// Guideline 16, R1), in the accumulators' own register order ([MT * NT][NTHR] 16-byte pieces: perfectly coalesced, and the reader has the same order);
// every wave drains its stores, the workgroup meets, ONE lane takes the tile's ticket (relaxed, agent scope).  The workgroup that draws S - 1 is the reducer:
// it reads the other slices' partials with sc1 loads (L2-served, never a stale L1 line; valid because the producers stored sc1) and adds
//     p(0) + p(1) + ... + p(S - 1)      left to right, its own partial from registers at its own position,
// so the result does not depend on WHICH slice arrived last -- bit-reproducible, no atomics on data.  It re-arms the ticket for the next launch (a recorded
// launch plan needs no memset node) and goes on to the epilogue.  No workgroup ever waits for another: nothing here depends on dispatch order or residency.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int S, int ME, int MT, int NT, int NTHR, int TILE_ELEMS>
TOC3D_DEV void sk_combine(f32x4 (&acc)[MT][NT], const __amdgpu_buffer_rsrc_t rs, const int tid) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4 v[S];
#pragma unroll
            for (int q = 0; q < S; ++q) {
                if (q == ME) v[q] = acc[i][j];
                else v[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (q * TILE_ELEMS + ((i * NT + j) * NTHR + tid) * 4) * 4, 0, 16));
            }
            f32x4 r = v[0];
#pragma unroll
            for (int q = 1; q < S; ++q) r += v[q];
            acc[i][j] = r;
        }
}
template <int MT, int NT, int NTHR, int TILE_ELEMS>
TOC3D_DEV bool sk_exchange(const GemmArgs& a, f32x4 (&acc)[MT][NT], char* smem, const int tile, const int slice, const int tid) {
    static_assert(MT * NT * NTHR * 4 == TILE_ELEMS, "the partial tile is the accumulators of the whole workgroup");
    const int S = a.split;
    // (a buffer descriptor on the tile's S partials: wave-uniform base, 32-bit offsets)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.sk_slabs + (size_t)tile * S * TILE_ELEMS, 0, S * TILE_ELEMS * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][j]), rs, (slice * TILE_ELEMS + ((i * NT + j) * NTHR + tid) * 4) * 4, 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY storing wave drains its own stores ...
    tile_barrier();                                          // ... before the workgroup meets (also: every wave is done with the operand stages, smem is free)
    unsigned* s_ticket = reinterpret_cast<unsigned*>(smem);
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(a.sk_tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)(S - 1)) __hip_atomic_store(a.sk_tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // last arriver: re-armed for the next launch
        *s_ticket = t;
    }
    lds_barrier();
    const unsigned ticket = *s_ticket;
    tile_barrier();                                          // every wave has read the word: smem is free again (the statistics epilogues overlay it)
    if (ticket != (unsigned)(S - 1)) return false;
    // all control of the sum is compile-time: S - 1 loads per accumulator and no branch around any of them (a run-time "register or load" select would
    // make hipcc wait per element: cdna_hip_programming.md, traps (c))
    switch (S * 8 + slice) {
        case 2 * 8 + 0: sk_combine<2, 0, MT, NT, NTHR, TILE_ELEMS>(acc, rs, tid); break;
        case 2 * 8 + 1: sk_combine<2, 1, MT, NT, NTHR, TILE_ELEMS>(acc, rs, tid); break;
        case 3 * 8 + 0: sk_combine<3, 0, MT, NT, NTHR, TILE_ELEMS>(acc, rs, tid); break;
        case 3 * 8 + 1: sk_combine<3, 1, MT, NT, NTHR, TILE_ELEMS>(acc, rs, tid); break;
        case 3 * 8 + 2: sk_combine<3, 2, MT, NT, NTHR, TILE_ELEMS>(acc, rs, tid); break;
        case 4 * 8 + 0: sk_combine<4, 0, MT, NT, NTHR, TILE_ELEMS>(acc, rs, tid); break;
        case 4 * 8 + 1: sk_combine<4, 1, MT, NT, NTHR, TILE_ELEMS>(acc, rs, tid); break;
        case 4 * 8 + 2: sk_combine<4, 2, MT, NT, NTHR, TILE_ELEMS>(acc, rs, tid); break;
        default: sk_combine<4, 3, MT, NT, NTHR, TILE_ELEMS>(acc, rs, tid); break;
    }
    return true;
}

// One BM x BN output tile (rows m0.., columns n0..) by the calling workgroup of 64 * WM * WN threads: K loop + fused epilogue.  `smem` = the
// workgroup's dynamic LDS.  gemm_kernel below runs one tile per workgroup.
// SK = 1: deterministic split-K (a.split workgroups per tile).  The calling workgroup multiplies K range `sk_slice` of tile `sk_tile` only; the partial
// accumulators meet through a.sk_slabs and the workgroup that arrives LAST at the tile's ticket adds them in slice order (so the sum does not depend on which
// one that is) and runs the epilogue; the others return after their store.  See sk_exchange below.
template <typename T, int EPI, int BM, int BN, int STAGES, int RB, int WM, int WN, int X3 = 0, int OCC = 1, int SK = 0>
TOC3D_DEV void gemm_tile(const GemmArgs& a, const int m0, const int n0, char* smem, const int sk_tile = 0, const int sk_slice = 0) {
    static_assert(SK == 0 || (EPI != TOC3D_EPI_CONV3X3 && !epi_is_rope(EPI)), "split-K serves the plain linear epilogues");
    static_assert(X3 == 0 || ((X3 == 3 || X3 == 6) && sizeof(T) == 4) || (X3 == 1 && sizeof(T) == 2),
                  "the bf16 x 3 / x 6 product forms run on f32 operands, the register-pipelined loop on bf16");
    constexpr bool PIPE = X3 == 1;                      // fragments of the NEXT 32-deep K step are read while the MFMAs of the current one run (see the loop)
    constexpr int NTHR = 64 * WM * WN;                  // WM x WN wavefronts
    constexpr int TM = BM / WM, TN = BN / WN;           // per-wave output tile
    constexpr int MT = TM / 16, NT = TN / 16;           // 16x16 MFMA tiles per wave
    constexpr int A_BYTES = BM * RB, STAGE_BYTES = (BM + BN) * RB;
    constexpr int LOADS = (BM * (RB / 16) + NTHR - 1) / NTHR + (BN * (RB / 16) + NTHR - 1) / NTHR;   // global_load_lds per thread per K-tile (stage_tile rounds up)
    // tiles with a partial DMA round (96- / 160-row tiles): the plain operand loaders, and -- round 6 -- the bf16 x 3 form when BOTH operands arrive as (hi, lo) planes
    // (nothing is split in LDS then: split_rows_x3's piece bookkeeping, which assumes whole rounds, is never used; launch_cfg refuses the other cases at run time)
    constexpr bool PARTIAL_ROUND = (BM * (RB / 16)) % NTHR != 0 || (BN * (RB / 16)) % NTHR != 0;
    static_assert(!PARTIAL_ROUND || ((X3 == 0 || X3 == 3) && EPI != TOC3D_EPI_CONV3X3), "tiles with a partial DMA round: plain operand loaders, or x3 on planes");
    constexpr int KS = RB / 32 / (int)sizeof(T);        // 32-wide K steps per K-tile
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, g = lane >> 4;

    f32x2* lnrow = reinterpret_cast<f32x2*>(smem + STAGES * STAGE_BYTES);   // [BM], only allocated for EPI_RESIDUAL_LN
    auto ln_rows_prepare = [&]() { ln_rows_prepare_fn<EPI, BM, NTHR>(a, m0, lnrow, tid); };
    // Register-capped tiles: the table is formed FIRST, while nothing else of the tile is live -- behind the first operand request (where the uncapped tiles do
    // it, overlapped with that request's flight) its batch of loads spills.  One exposed L2 round trip instead of 4-11 dependent ones.
    constexpr bool PREP_EARLY = epi_ln_stats_in(EPI) && OCC > 1;
    if constexpr (PREP_EARLY) {
        ln_rows_prepare();
        __builtin_amdgcn_sched_barrier(0);
    }

    const T* A = reinterpret_cast<const T*>(a.A);
    const T* W = reinterpret_cast<const T*>(a.W);
    constexpr int BK = RB / (int)sizeof(T);
    // split-K: slice q of S multiplies elements [b(q), b(q + 1)) of K, b(q) = 128 * round(q * K / (128 S)) -- whole K-tiles of every variant (BK divides 128), and
    // the same cut for every variant, so the bits depend on S alone (one bit class per S, like the unsplit variants among themselves)
    static_assert(SK == 0 || 128 % BK == 0, "split-K cuts K at multiples of 128 elements");
    int k_base = 0, nk = a.K / BK;
    if constexpr (SK != 0) {
        const int n64 = a.K / 64, S = a.split;
        const int lo = sk_slice == 0 ? 0 : 128 * ((sk_slice * n64 + S) / (2 * S)), hi = sk_slice + 1 == S ? a.K : 128 * (((sk_slice + 1) * n64 + S) / (2 * S));
        k_base = lo;
        nk = (hi - lo) / BK;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // W rows are padded to a multiple of 128 at pack time, A rows are clamped to M-1
    const int w_max = ((a.N + 127) / 128) * 128 - 1;
    // EPI_CONV3X3 (necks/cp_fpn.py:124-133 as an implicit GEMM): the A tile of K-tile t is the (ky, kx) = tap t*BK / C neighbour of each
    // row's pixel, channels (t*BK) % C ..; the 16-byte global_load_lds takes any per-lane source address, so the im2col matrix is never
    // materialised -- out-of-image taps read a line of zeros.  Each thread's rows are fixed: their (y, x) are decoded once.
    constexpr int LA = BM * (RB / 16) / NTHR;
    int cv_m[EPI == TOC3D_EPI_CONV3X3 ? LA : 1], cv_yx[EPI == TOC3D_EPI_CONV3X3 ? LA : 1];
    if constexpr (EPI == TOC3D_EPI_CONV3X3) {
#pragma unroll
        for (int t = 0; t < LA; ++t) {
            const int r = (t * NTHR + wave * 64 + lane) / (RB / 16);
            int m = m0 + r;
            m = m < a.M ? m : a.M - 1;
            cv_m[t] = m;
            cv_yx[t] = (((m / a.conv_w) % a.conv_h) << 16) | (m % a.conv_w);
        }
    }
    auto request = [&](int t) {
        char* slot = smem + (t % STAGES) * STAGE_BYTES;
        if constexpr (EPI == TOC3D_EPI_CONV3X3) {
            const int k0 = t * BK, C = (int)a.lda;
            const int tap = k0 / C, c0 = k0 - tap * C;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
            for (int u = 0; u < LA; ++u) {
                const int cidx = u * NTHR + wave * 64 + lane;
                const int r = cidx / (RB / 16), pch = cidx % (RB / 16);
                const int y2 = (cv_yx[u] >> 16) + dy, x2 = (cv_yx[u] & 0xffff) + dx;
                const bool in = (unsigned)y2 < (unsigned)a.conv_h && (unsigned)x2 < (unsigned)a.conv_w;
                // out-of-image taps: any 16 zero bytes (no chunk offset: K-tiles of 256 / 512 bytes would run past a small zero line)
                const char* src = in ? reinterpret_cast<const char*>(A + (int64_t)(cv_m[u] + dy * a.conv_w + dx) * C + c0) + ((pch ^ swz<RB>(r)) << 4)
                                     : reinterpret_cast<const char*>(a.zeros);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(slot + (u * NTHR + wave * 64) * 16), 16, 0, 0);
            }
        } else {
            stage_tile<T, BM, RB, NTHR>(A, a.lda, m0, a.M - 1, k_base + t * BK, slot, wave, lane);
        }
#ifndef TOC3D_W_AUX
#define TOC3D_W_AUX 0                                   // experiment (round 5, profiles/r05_nt_stores.txt): cache policy bits of the W operand's DMA loads (2 = nt: stream past the L2's LRU)
#endif
        stage_tile<T, BN, RB, NTHR, TOC3D_W_AUX>(W, a.ldw, n0, w_max, k_base + t * BK, slot + A_BYTES, wave, lane);
    };
    auto multiply = [&](int t) {
        const char* sA = smem + (t % STAGES) * STAGE_BYTES;
        const char* sB = sA + A_BYTES;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Frag<T> fa[MT], fb[NT];
            if constexpr (X3 != 3) {
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[i] = lds_frag<RB>(sA, wm * TM + i * 16 + r16, s, g, T());
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[j] = lds_frag<RB>(sB, wn * TN + j * 16 + r16, s, g, T());
            }
            if constexpr (X3 == 6) {
                bf16x8 ah[MT], am[MT], al[MT], bh[NT], bm[NT], bl[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) split_bf16x6(fa[i], ah[i], am[i], al[i]);
#pragma unroll
                for (int j = 0; j < NT; ++j) split_bf16x6(fb[j], bh[j], bm[j], bl[j]);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {           // smallest terms first
                        f32x4 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm[j], am[i], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm[j], ah[i], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], am[i], c, 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], c, 0, 0, 0);
                    }
            } else if constexpr (X3 == 3) {
                // the operand tiles were split in place by the wave that staged them (split_rows_x3 below): per 32-k group of a row, chunks 0-3 = hi,
                // chunks 4-7 = lo of the same 32 elements
                bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int r = wm * TM + i * 16 + r16;
                    ah[i] = *reinterpret_cast<const bf16x8*>(sA + r * RB + (((s * 8 + g) ^ swz<RB>(r)) << 4));
                    al[i] = *reinterpret_cast<const bf16x8*>(sA + r * RB + (((s * 8 + 4 + g) ^ swz<RB>(r)) << 4));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int r = wn * TN + j * 16 + r16;
                    bh[j] = *reinterpret_cast<const bf16x8*>(sB + r * RB + (((s * 8 + g) ^ swz<RB>(r)) << 4));
                    bl[j] = *reinterpret_cast<const bf16x8*>(sB + r * RB + (((s * 8 + 4 + g) ^ swz<RB>(r)) << 4));
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {           // small terms first; operands swapped like mma_step below
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                    }
            } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) mma_step(acc[i][j], fb[j], fa[i]);   // swapped: C^T tile layout, see the epilogue
            }
        }
    };
    // EPI_RESIDUAL_LN -- folded LayerNorm of the A rows: (mean, rstd) per tile row from the partial sums the producing GEMM left (include/toc3d.h),
    // into a row table behind the operand stages.  Done at kernel start, right after the first operand tiles were requested, so its one
    // global round trip overlaps theirs.  Four threads per row; thread part p sums slots p, p + 4, ... in sequence, the parts meet in two
    // butterfly steps; f64.  The order is fixed: the bits do not depend on the tile variant that runs this kernel.
    // bf16 x 3: the split of the f32 operands into (hi, lo) bf16 planes happens ONCE per element, in LDS, by the wave whose DMA instruction brought the
    // row (a global_load_lds instruction of stage_tile covers whole rows, all lanes of a row sit in one wavefront): after its own vmcnt wait every lane
    // converts the 16-byte chunk it staged -- 4 floats -> 4 hi + 4 lo bf16, split_bf16x3's arithmetic -- and stores them into the hi / lo chunk of the
    // row's 32-k group (8 bytes each; reads precede writes inside the wave, so the conversion is in place).  The first form split every fragment in
    // registers in every wave that multiplied it: 204 VALU instructions per 24 MFMAs in the K loop (profiles/r03_x3_split.txt).
    auto split_rows_x3 = [&](int t) {
        if constexpr (X3 == 3 && !PARTIAL_ROUND) {
            char* slot = smem + (t % STAGES) * STAGE_BYTES;
            constexpr int CPR = RB / 16, PER = (BM + BN) * CPR / NTHR;      // the A tile and the W tile are contiguous: rows 0 .. BM + BN - 1 of RB bytes
            f32x4 v[PER];
            int pos[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                // stage_tile's piece order: A pieces first (BM * CPR / NTHR per thread), then W pieces, each t * NTHR + wave * 64 + lane
                constexpr int PA = BM * CPR / NTHR;
                const int cidx = (u < PA ? u : u - PA) * NTHR + wave * 64 + lane;
                pos[u] = (u < PA ? 0 : BM * CPR) + cidx;                    // chunk position in the stage (row * CPR + stored chunk)
                if (u < PA ? a.a_planes : a.w_planes) continue;             // the operand came as planes (wave-uniform): nothing to split
                v[u] = *reinterpret_cast<const f32x4*>(slot + pos[u] * 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                if (u < BM * CPR / NTHR ? a.a_planes : a.w_planes) continue;
                const int row = pos[u] / CPR, p = pos[u] % CPR;
                const int rt = row < BM ? row : row - BM;                  // tile-local row: the swizzle of stage_tile
                const int sw = swz<RB>(rt);
                const int c = p ^ sw;                                       // logical f32 chunk of the row held at stored position p
                const int grp = c >> 3, cc = c & 7;                        // 32-k group, chunk inside it (4 floats)
                typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
                bf16x4_t hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bf16_t h = (bf16_t)v[u][e];
                    hi[e] = h;
                    lo[e] = (bf16_t)(v[u][e] - (float)h);
                }
                char* rowp = slot + (pos[u] - p) * 16;
                *reinterpret_cast<bf16x4_t*>(rowp + (((grp * 8 + (cc >> 1)) ^ sw) << 4) + (cc & 1) * 8) = hi;
                *reinterpret_cast<bf16x4_t*>(rowp + (((grp * 8 + 4 + (cc >> 1)) ^ sw) << 4) + (cc & 1) * 8) = lo;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    // (Round 4, measured and removed: the statistics rows brought to LDS by DMA beside the first operand tile instead of through registers.  No gain on any tile --
    // with the table stubbed out the whole cost of this epilogue disappears, with the rows in LDS it stays: what costs is the f64 reduction between the first
    // barrier and the first MFMA, paid by every one of the 43 column tiles of a row panel, not the loads -- and the extra LDS and registers halved the 4-wave tile.
    // profiles/r04_fold_epilogue_cost.txt.)
    // EPI_QKV_ROPE: every lane fetches the RoPE positions of its MT output rows at kernel start, and the compact tables ([cos | sin], 4-5 KB)
    // come to LDS by DMA while the K loop runs, so that the epilogue has no dependent global round trip (the first version read both from
    // global memory inside the epilogue: +7 us per q|k|v launch).  Single-buffer tiles: a region behind the operand stage, requested with the
    // first K-tile (those tiles leave LDS to spare).  Rings: the slot that the last K-tile's iteration would otherwise refill -- no extra LDS,
    // a second 80 KB workgroup still fits the CU -- requested in front of the last multiply.
    const float* rope_tab = nullptr;
    int rope_rcs[epi_is_rope(EPI) ? MT : 1];
    auto rope_stage = [&](char* dst) {
        const int nchunk = a.rope_L * 16;                // 16-byte pieces of [cos | sin]
        for (int c0 = wave * 64; c0 < nchunk; c0 += NTHR) {
            int c = c0 + lane;
            c = c < nchunk ? c : 0;                      // the last instruction's surplus lanes re-read piece 0 into the slack behind the table
            __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(a.rope_tab) + c * 16), (lptr_t)(dst + c0 * 16), 16, 0, 0);
        }
        rope_tab = reinterpret_cast<const float*>(dst);
    };
    if constexpr (epi_is_rope(EPI)) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = m0 + wm * TM + i * 16 + r16;
            rope_rcs[i] = a.rope_rc[row < a.M ? row : a.M - 1];
        }
    }
    if constexpr (PIPE) {
        // Register-pipelined ring (round 5) for the launches that run ONE workgroup per CU (N = 1024 at M < 6000: 176-392 tiles).  The loop below this one reads a K step's
        // fragments and then multiplies them, every wave in step between two barriers: with no second workgroup on the CU the matrix pipe idles through every LDS round trip
        // (deep rings at one workgroup per CU ran at half the rate of single buffers at four, LABNOTES.md).  Here a wave holds TWO fragment sets: the reads of K step
        // (kt, 1) are issued in front of the MFMAs of (kt, 0), those of (kt + 1, 0) in front of the MFMAs of (kt, 1) -- the barrier that publishes tile kt + 1 therefore sits in
        // the MIDDLE of K-tile kt, half a tile earlier than in the plain ring (one tile less may stay in flight: STAGES >= 3, meant for 4).
        //   RAW: tile kt + 1 is read behind the counted wait + barrier of K-tile kt.
        //   WAR: behind that barrier the slot of tile kt - 1 is refilled; its last fragments (kt - 1, 1) were consumed by MFMAs every wave issued before arriving.
        static_assert(STAGES >= 3 && KS == 2 && EPI != TOC3D_EPI_CONV3X3, "register-pipelined loop: rings of >= 3 K-tiles of 64 bf16, linear epilogues");
        Frag<T> fa0[MT], fb0[NT], fa1[MT], fb1[NT];
        auto read_set = [&](int t, auto S, Frag<T> (&fa)[MT], Frag<T> (&fb)[NT]) {
            constexpr int sidx = decltype(S)::value;
            const char* sA = smem + (t % STAGES) * STAGE_BYTES;
            const char* sB = sA + A_BYTES;
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = lds_frag<RB>(sA, wm * TM + i * 16 + r16, sidx, g, T());
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = lds_frag<RB>(sB, wn * TN + j * 16 + r16, sidx, g, T());
        };
        auto mfma_set = [&](auto FIRST, const Frag<T> (&fa)[MT], const Frag<T> (&fb)[NT]) {      // FIRST = 1: MFMA (0, 0) only; 0: the others; 2: all
            constexpr int first = decltype(FIRST)::value;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (first == 2 || (first == 1) == (i == 0 && j == 0)) mma_step(acc[i][j], fb[j], fa[i]);   // swapped: C^T tile layout, see the epilogue
        };
        using F0 = std::integral_constant<int, 0>;
        using F1 = std::integral_constant<int, 1>;
        using F2 = std::integral_constant<int, 2>;
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            if (t < nk) request(t);
        if constexpr (epi_ln_stats_in(EPI) && !PREP_EARLY) ln_rows_prepare();
        if (nk - 1 >= STAGES - 2) wait_vmcnt<(STAGES - 2) * LOADS>();                  // tile 0 landed; tiles 1 .. STAGES - 2 may stay in flight
        else wait_vmcnt<0>();
        tile_barrier();
        read_set(0, S0(), fa0, fb0);
        for (int kt = 0; kt < nk; ++kt) {
            // fa0 / fb0 were requested in the previous iteration: hipcc's wait in front of their first use is lgkmcnt(0) (it does not order LDS reads across the loop's back
            // edge), so that first MFMA goes IN FRONT of this iteration's reads -- behind them the wait would cover the new reads too and expose them
            mfma_set(F1(), fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            read_set(kt, S1(), fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_set(F0(), fa0, fb0);
            if (kt + 1 < nk) {
                if (nk - 2 - kt >= STAGES - 3) wait_vmcnt<(STAGES - 3) * LOADS>();      // tile kt + 1 landed; tiles kt + 2 .. kt + STAGES - 2 may stay in flight
                else wait_vmcnt<0>();                                                   // pipeline tail
            }
            tile_barrier();
            if (kt + STAGES - 1 < nk) request(kt + STAGES - 1);                         // into the slot of tile kt - 1
            if constexpr (epi_is_rope(EPI)) { if (kt == nk - 1) rope_stage(smem + ((kt + STAGES - 1) % STAGES) * STAGE_BYTES); }
            // (the same order again: across the branches above hipcc's wait for fa1 / fb1 is lgkmcnt(0) too.  The read is unconditional -- the last iteration reads a stale
            // slot, unused -- because behind a branch the reads would sit in a block of their own and the scheduler fences below would not hold them)
            mfma_set(F1(), fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            read_set(kt + 1, S0(), fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_set(F0(), fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (epi_is_rope(EPI)) { wait_vmcnt<0>(); tile_barrier(); }
    } else if (STAGES == 1) {
        // single LDS buffer, two barriers per K-tile; latency is hidden by co-resident workgroups (small LDS footprint)
        for (int kt = 0; kt < nk; ++kt) {
            request(kt);
            if constexpr (epi_ln_stats_in(EPI) && !PREP_EARLY) { if (kt == 0) ln_rows_prepare(); }
            if constexpr (epi_is_rope(EPI)) { if (kt == 0) rope_stage(smem + STAGE_BYTES + (epi_ln_in(EPI) ? BM * 8 : 0)); }
            wait_vmcnt<0>();
            split_rows_x3(kt);
            tile_barrier();                              // every wave's pieces of tile kt have landed
            multiply(kt);
            tile_barrier();                              // every wave is done reading: the buffer may be overwritten
        }
    } else {
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            if (t < nk) request(t);
        if constexpr (epi_ln_stats_in(EPI) && !PREP_EARLY) ln_rows_prepare();
        for (int kt = 0; kt < nk; ++kt) {
            if (nk - 1 - kt >= STAGES - 2) wait_vmcnt<(STAGES >= 2 ? STAGES - 2 : 0) * LOADS>();   // tiles kt+1 .. kt+STAGES-2 may stay in flight
            else wait_vmcnt<0>();                                                                   // pipeline tail
            split_rows_x3(kt);
            tile_barrier();
            if (kt + STAGES - 1 < nk) request(kt + STAGES - 1);
            if constexpr (epi_is_rope(EPI)) { if (kt == nk - 1) rope_stage(smem + ((kt + STAGES - 1) % STAGES) * STAGE_BYTES); }   // the slot tile kt - 1 left
            multiply(kt);
        }
        if constexpr (epi_is_rope(EPI)) { wait_vmcnt<0>(); tile_barrier(); }
    }
    TOC3D_TRACE(1);
    if constexpr (SK != 0) {
        if (!sk_exchange<MT, NT, NTHR, BM * BN>(a, acc, smem, sk_tile, sk_slice, tid)) return;      // not the last arriver of this tile: the partial is stored, done
    }
    tile_finish<T, EPI, BM, BN, WM, WN, (STAGES > 1)>(a, acc, m0, n0, smem, lnrow, rope_rcs, rope_tab);
}

template <typename T, int EPI, int BM, int BN, int STAGES, int RB, int WM, int WN, int OCC, int X3 = 0, int SK = 0>
__global__ __launch_bounds__(64 * WM * WN, OCC) void gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TOC3D_TRACE(0);
    const int tiles_n = (a.N + BN - 1) / BN;
    const int tiles_m = (a.M + BM - 1) / BM;
    int m0, n0;
    if constexpr (SK != 0) {
        // split-K: grid = tiles * split.  Unit u = slice * tiles + tile, the units cut into 8 contiguous XCD chunks: an XCD works on neighbouring tiles of ONE
        // K range (their A row panels and that range of W stay in its L2), and the slices of a tile run on different XCDs at about the same time.
        const int tiles = tiles_m * tiles_n;
        const int u = xcd_remap(blockIdx.x, tiles * a.split);
        const int slice = u / tiles, tile = u - slice * tiles;
        gemm_tile<T, EPI, BM, BN, STAGES, RB, WM, WN, X3, OCC, 1>(a, (tile / tiles_n) * BM, (tile % tiles_n) * BN, smem, tile, slice);
        TOC3D_TRACE_END();
        return;
    }
    if (a.order == 0) {
        const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
        m0 = (tile / tiles_n) * BM;
        n0 = (tile % tiles_n) * BN;
    } else if (a.order == 1) {
        // Workgroup b runs on XCD b % 8 (observed dispatch rule, used for speed only).  Each XCD owns a band of M-tile rows
        // whose A panels (band * K bytes, ~1.5 MB) stay resident in its 4 MB L2, and walks the W panels one after the
        // other (m fastest), so every W panel is fetched from memory once per XCD instead of once per A row-panel.
        const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        const int r0 = (xcd * tiles_m) >> 3, r1 = ((xcd + 1) * tiles_m) >> 3;
        const int band = r1 - r0;
        if (band <= 0 || l >= band * tiles_n) return;
        m0 = (r0 + l % band) * BM;
        n0 = (l / band) * BN;
    } else {
        // 2-D XCD partition for wide N (order 2: 4 row bands x 2 column halves, order 3: 2 x 4): with row bands alone every XCD streams ALL of W
        // (w1|w2: 11 MB x 8 XCDs per launch through the fabric); here an XCD keeps a quarter (half) of the A rows resident and streams half (a
        // quarter) of W, m fastest inside its block.
        const int pm = a.order == 2 ? 4 : 2, pn = 8 / pm;
        const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        const int bm = xcd / pn, bn = xcd % pn;
        const int r0 = (bm * tiles_m) / pm, r1 = ((bm + 1) * tiles_m) / pm;
        const int c0 = (bn * tiles_n) / pn, c1 = ((bn + 1) * tiles_n) / pn;
        const int hm = r1 - r0, hn = c1 - c0;
        if (hm <= 0 || hn <= 0 || l >= hm * hn) return;
        m0 = (r0 + l % hm) * BM;
        n0 = (c0 + l / hm) * BN;
    }
    gemm_tile<T, EPI, BM, BN, STAGES, RB, WM, WN, X3, OCC>(a, m0, n0, smem);
    TOC3D_TRACE_END();
}

// ---------------------------------------------------------------------------------------------------
// Phased big-tile kernel (bf16): BM x BN per 512-thread workgroup, ONE workgroup per CU, latency hidden inside the workgroup.
//
// The 128x128 family above is bound by the L2 -> LDS fill rate (32 KB per 2.1 MFLOP K-step, LABNOTES.md) and hides
// latency only through co-resident workgroups.  Here a K-step of a 256x256 tile brings 64 KB for 8.4 MFLOP (half the bytes per
// FLOP), every wavefront owns a 128x64 block (half the LDS read bytes per FLOP of the 64x32 blocks above), and the K loop is
// cut into four *phases* per 64-deep K-tile (cdna_hip_programming.md "8-phase" template, re-derived for this ring):
//
//   the wave's block = 2 x 2 sub-blocks (a0 | a1 rows) x (b0 | b1 columns); the LDS ring holds two K-tiles, each as four
//   half-tiles  A0 A1 B0 B1  (A_h = the a_h rows of every wave row, B_h likewise);
//   phase 1: read A0[t]    | stage A1[t+1] | MFMA a0.b0          phase 3: read A1[t]    | stage A0[t+2]       | MFMA a1.b1
//   phase 2: read B1[t]    | stage B0[t+2] | MFMA a0.b1          phase 4: read B0[t+1]  | stage B1[t+2], wait | MFMA a1.b0
//   (two B fragment sets that swap roles every K-tile: no half-tile is read twice, round 5; hazards at `ktile` in the kernel)
//
//   each phase = [LDS reads + one half-tile of global_load_lds] s_barrier [MFMAs] s_barrier, and the second half of the
//   wavefronts (waves 4-7, which share the SIMDs of waves 0-3) runs ONE barrier behind the first: on every SIMD one wave is in its
//   MFMA segment while its partner issues loads, so the matrix pipe and the memory path stay busy from a single workgroup.
//   LDS budget of a 256x256 K-tile at 128 B / clock: 8 waves x 24 KB of fragment reads + 64 KB of DMA writes = 2048 clocks = the MFMA time of the
//   K-tile (8.4 MFLOP at 4096 FLOP / clock / CU): the LDS port is co-critical, which is what the re-read of B0 (28 KB per wave) was costing.
// ---------------------------------------------------------------------------------------------------
template <int ROWS, int TB, int NTHR>
TOC3D_DEV void stage_half(const char* __restrict__ g, int64_t ld_bytes, int row0, int max_row, int64_t k0_bytes, int h, char* lds_half, int wave, int lane) {
    // half-tile h of an operand whose wave blocks are TB rows tall: LDS row lr = (wave-row w) * TB/2 + j  <->  tile row w * TB + h * TB/2 + j
    constexpr int L = ROWS * 8 / NTHR;
    static_assert(L * NTHR == ROWS * 8, "half-tile must be a whole number of 16-byte loads per thread");
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int c = i * NTHR + wave * 64 + lane;
        const int lr = c >> 3, p = c & 7;
        const int w = lr / (TB / 2), j = lr % (TB / 2);
        int gr = row0 + w * TB + h * (TB / 2) + j;
        gr = gr < max_row ? gr : max_row;
        const char* src = g + (int64_t)gr * ld_bytes + k0_bytes + ((p ^ (lr & 7)) << 4);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_half + (i * NTHR + wave * 64) * 16), 16, 0, 0);
    }
}

// X3 (round 6): the bf16 x 3 form on (hi, lo) PLANES (TOC3D_DTYPE_F32X3P: both operands in planes).  A 128-byte row piece of a planes operand is [32 x hi | 32 x lo] of
// 32 consecutive k -- byte for byte the image the bf16 kernel stages for 64 consecutive k -- so staging, LDS layout and fragment reads are unchanged: a K-tile covers 32 k
// (nk = K / 32), the fragment of "K step 0" is hi and that of "K step 1" lo, and a (row tile, column tile) pair takes THREE MFMAs per K-tile (lo.hi, hi.lo, hi.hi: the order
// of gemm_tile's X3 loop, so every x3 variant returns the same bits) instead of two: 1.5x the matrix-core work per LDS byte of the bf16 form, whose LDS port is co-critical.
template <int EPI, int BM, int BN, int WM, int WN, bool X3 = false>
__global__ __launch_bounds__(512, 2) void gemm_phased_kernel(GemmArgs a) {
    using T = std::conditional_t<X3, float, bf16_t>;      // element type of the epilogue's buffers
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(WM * WN == 8, "eight wavefronts");
    constexpr int NTHR = 512;
    constexpr int TM = BM / WM, TN = BN / WN;             // per-wave block
    constexpr int HM = TM / 2, HN = TN / 2;               // sub-blocks
    constexpr int MT2 = HM / 16, NT2 = HN / 16;           // MFMA tiles per sub-block
    static_assert(HM % 16 == 0 && HN % 16 == 0, "sub-blocks are whole MFMA tiles");
    constexpr int AH = (BM / 2) * 128, BH = (BN / 2) * 128;   // bytes per half-tile (128-byte rows: 64 bf16 of K)
    constexpr int KT = 2 * AH + 2 * BH;                   // one K-tile in the ring: A0 | A1 | B0 | B1
    constexpr int LA = (BM / 2) * 8 / NTHR, LB = (BN / 2) * 8 / NTHR;   // global_load_lds per thread per half-tile
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, g = lane >> 4;
    const bool late = wave >= 4;                          // the group that runs one barrier behind
    TOC3D_TRACE(0);

    const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    int m0, n0;
    if (a.order == 0) {
        const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
        m0 = (tile / tiles_n) * BM;
        n0 = (tile % tiles_n) * BN;
    } else {
        const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        const int r0 = (xcd * tiles_m) >> 3, r1 = ((xcd + 1) * tiles_m) >> 3;
        const int band = r1 - r0;
        if (band <= 0 || l >= band * tiles_n) return;
        m0 = (r0 + l % band) * BM;
        n0 = (l / band) * BN;
    }
    const char* A = reinterpret_cast<const char*>(a.A);
    const char* W = reinterpret_cast<const char*>(a.W);
    const int64_t lda_b = a.lda * (int64_t)sizeof(T), ldw_b = a.ldw * (int64_t)sizeof(T);
    const int nk = a.K / (X3 ? 32 : 64);                  // 128 bytes of every row per K-tile
    const int a_max = a.M - 1, w_max = ((a.N + 127) / 128) * 128 - 1;
    // behind the two-K-tile ring: the (mean, rstd) row table of the LayerNorm-consuming epilogues, then the RoPE tables of the rotating q|k|v epilogue
    f32x2* lnrow = reinterpret_cast<f32x2*>(smem + 2 * KT);
    const float* rope_tab = nullptr;
    int rope_rcs[epi_is_rope(EPI) ? 2 * MT2 : 1];
    if constexpr (epi_is_rope(EPI)) {
        // requested FIRST: the oldest requests of the wave, so every counted s_waitcnt vmcnt of the K loop has retired them long before the epilogue
        char* dst = smem + 2 * KT + (epi_ln_in(EPI) ? BM * 8 : 0);
        const int nchunk = a.rope_L * 16;                // 16-byte pieces of [cos | sin]
        for (int c0 = wave * 64; c0 < nchunk; c0 += NTHR) {
            int c = c0 + lane;
            c = c < nchunk ? c : 0;
            __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(a.rope_tab) + c * 16), (lptr_t)(dst + c0 * 16), 16, 0, 0);
        }
        rope_tab = reinterpret_cast<const float*>(dst);
#pragma unroll
        for (int i = 0; i < 2 * MT2; ++i) {
            const int row = m0 + wm * TM + i * 16 + r16;
            rope_rcs[i] = a.rope_rc[row < a.M ? row : a.M - 1];
        }
    }
    // the row table before anything else of the tile is live (one exposed L2 round trip; behind the prologue's requests its loads would drain the whole
    // DMA queue: hipcc waits vmcnt(0) for an ordinary load issued beside LDS-DMA)
    if constexpr (epi_ln_stats_in(EPI)) {
        ln_rows_prepare_fn<EPI, BM, NTHR>(a, m0, lnrow, tid);
        __builtin_amdgcn_sched_barrier(0);
    }

    f32x4 acc[2 * MT2][2 * NT2];
#pragma unroll
    for (int i = 0; i < 2 * MT2; ++i)
#pragma unroll
        for (int j = 0; j < 2 * NT2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frag<bf16_t> fa[MT2][2], fb0[NT2][2], fb1[NT2][2];    // [tile][32-wide K step]; the two B sets swap roles every K-tile (see ktile)

    auto slot = [&](int t, int kind, int h) -> char* { return smem + (t & 1) * KT + kind * 2 * AH + h * (kind ? BH : AH); };
    auto stage_a = [&](int t, int h) { stage_half<BM / 2, TM, NTHR>(A, lda_b, m0, a_max, (int64_t)t * 128, h, slot(t, 0, h), wave, lane); };
    auto stage_b = [&](int t, int h) { stage_half<BN / 2, TN, NTHR>(W, ldw_b, n0, w_max, (int64_t)t * 128, h, slot(t, 1, h), wave, lane); };
    auto read_a = [&](int t, int h) {
        const char* base = slot(t, 0, h);
#pragma unroll
        for (int i = 0; i < MT2; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[i][ks] = lds_frag<128>(base, wm * HM + i * 16 + r16, ks, g, bf16_t());
    };
    auto read_b = [&](int t, int h, Frag<bf16_t> (&fb)[NT2][2]) {
        const char* base = slot(t, 1, h);
#pragma unroll
        for (int j = 0; j < NT2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[j][ks] = lds_frag<128>(base, wn * HN + j * 16 + r16, ks, g, bf16_t());
    };
    auto mfma = [&](auto HA, auto HB, const Frag<bf16_t> (&fb)[NT2][2]) {
        constexpr int ha = decltype(HA)::value, hb = decltype(HB)::value;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (X3) {
#pragma unroll
            for (int i = 0; i < MT2; ++i)
#pragma unroll
                for (int j = 0; j < NT2; ++j) {      // [0] = hi, [1] = lo; small terms first, the order of gemm_tile's X3 loop
                    f32x4& c = acc[ha * MT2 + i][hb * NT2 + j];
                    mma_step(c, fb[j][1], fa[i][0]);
                    mma_step(c, fb[j][0], fa[i][1]);
                    mma_step(c, fb[j][0], fa[i][0]);
                }
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MT2; ++i)
#pragma unroll
                for (int j = 0; j < NT2; ++j) mma_step(acc[ha * MT2 + i][hb * NT2 + j], fb[j][ks], fa[i][ks]);   // swapped: see gemm_epilogue
        }
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // One K-tile.  fbp holds B0[t] on entry (read during phase 4 of K-tile t - 1, or by the prologue); fbq takes B1[t] in phase 2 and, once phase 3's
    // MFMAs are done with it, B0[t + 1] in phase 4 -- so no half-tile is read twice (round 4 re-read B0 in phase 4) and the LDS reads per phase are
    // 2 MT2 | 2 NT2 | 2 MT2 | 2 NT2 fragments instead of 2 (MT2 + NT2) | 2 NT2 | 2 MT2 | 2 NT2: the first phase no longer carries half of the K-tile's reads.
    //   phase 1: read A0[t]      | stage A1[t+1] | MFMA a0.b0          phase 3: read A1[t]      | stage A0[t+2]       | MFMA a1.b1
    //   phase 2: read B1[t]      | stage B0[t+2] | MFMA a0.b1          phase 4: read B0[t+1]    | stage B1[t+2], wait | MFMA a1.b0, lgkmcnt(0)
    // RAW: the counted wait of phase 4 leaves only A0[t+2] / B1[t+2] in flight: A0, B1, A1 of K-tile t + 1 and B0[t+2] (issued in that order before them) have
    //      landed one barrier before their first read.  B0[t+1] itself was retired by the wait of K-tile t - 1 (or the prologue's).
    // WAR: A0 / B1 / A1 slots as before (restaged two phases or more after their read).  B0[t+1]'s slot is restaged in phase 2 of K-tile t + 1 with B0[t+3];
    //      its read is issued in phase 4 of K-tile t and consumed two barriers later, so the explicit lgkmcnt(0) that closes phase 4's MFMA segment is what
    //      puts the read's completion (both wave groups) in front of the barriers the restaging wave passes first.
    auto ktile = [&](int t, Frag<bf16_t> (&fbp)[NT2][2], Frag<bf16_t> (&fbq)[NT2][2]) {
        const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
        // phase 1
        read_a(t, 0);
        if (n1) stage_a(t + 1, 1);
        tile_barrier();
        mfma(I0(), I0(), fbp);
        tile_barrier();
        // phase 2
        read_b(t, 1, fbq);
        if (n2) stage_b(t + 2, 0);
        tile_barrier();
        mfma(I0(), I1(), fbq);
        tile_barrier();
        // phase 3
        read_a(t, 1);
        if (n2) stage_a(t + 2, 0);
        tile_barrier();
        mfma(I1(), I1(), fbq);
        tile_barrier();
        // phase 4
        if (n1) read_b(t + 1, 0, fbq);
        if (n2) { stage_b(t + 2, 1); wait_vmcnt<LA + LB>(); }
        else wait_vmcnt<0>();
        tile_barrier();
        mfma(I1(), I0(), fbp);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tile_barrier();
    };

    // ---- prologue: K-tile 0 and B0 of K-tile 1 complete, A0 / B1 of K-tile 1 on their way (what phases 2-4 of a tile "-1" would have staged) ----
    stage_a(0, 0); stage_b(0, 0); stage_b(0, 1); stage_a(0, 1);
    if (nk > 1) { stage_b(1, 0); stage_a(1, 0); stage_b(1, 1); wait_vmcnt<LA + LB>(); }
    else wait_vmcnt<0>();
    tile_barrier();
    if (late) tile_barrier();
    read_b(0, 0, fb0);

    for (int t = 0; t < nk; t += 2) {
        ktile(t, fb0, fb1);
        if (t + 1 < nk) ktile(t + 1, fb1, fb0);
    }
    if (!late) tile_barrier();                            // every wave executes the same number of barriers (and is done with the ring: the statistics overlay it)
    TOC3D_TRACE(1);

    tile_finish<T, EPI, BM, BN, WM, WN, false>(a, acc, m0, n0, smem, lnrow, rope_rcs, rope_tab);
    TOC3D_TRACE_END();
}


template <typename T, int EPI, int BM, int BN, int STAGES, int RB = 128, int WM = 2, int WN = 2, int OCC = 1, int X3 = 0>
void launch_cfg(const GemmArgs& a, hipStream_t s) {
    // a wave must own whole (w1, w2) 32-column groups; the folded-LayerNorm statistics need N-tiles of whole 128-column slots; the fold is bf16 only
    constexpr bool unsupported = (epi_is_swiglu(EPI) && (BN / WN) % 32 != 0) || (epi_stats_out(EPI) && BN % (epi_is_swiglu(EPI) ? 128 : 64) != 0) ||
                                 (EPI >= TOC3D_EPI_SWIGLU_STATS && EPI != TOC3D_EPI_CONV3X3 && sizeof(T) != 2 &&
                                  !(X3 == 3 && (EPI == TOC3D_EPI_SWIGLU_STATS || EPI == TOC3D_EPI_RESIDUAL_LN || EPI == TOC3D_EPI_RESIDUAL_STATS ||
                                                EPI == TOC3D_EPI_SWIGLU_STATS_LN || EPI == TOC3D_EPI_QKV_ROPE)));   // ... and the bf16 x 3 forms of the ffn_ln and norm2 folds (f32 copies, f32 statistics)
    constexpr bool partial_round = (BM * (RB / 16)) % (64 * WM * WN) != 0 || (BN * (RB / 16)) % (64 * WM * WN) != 0;      // 96- / 160-row tiles
    if constexpr (unsupported || (partial_round && ((X3 != 0 && X3 != 3) || EPI == TOC3D_EPI_CONV3X3))) {
        g_bad_variant = true;
    } else {
        if (partial_round && X3 == 3 && !(a.a_planes && a.w_planes)) { g_bad_variant = true; return; }      // x3 on a 96- / 160-row tile: both operands must be planes (gemm_tile)
        constexpr int lds_fixed = STAGES * (BM + BN) * RB + (epi_ln_in(EPI) ? BM * 8 : 0);   // + the (mean, rstd) row table
        const int lds = lds_fixed + (epi_is_rope(EPI) && STAGES == 1 ? (a.rope_L * 256 + 1023) / 1024 * 1024 : 0);   // single buffer: + the RoPE tables (cos | sin), whole DMA instructions
        static Toc3dLdsAttr attr;          // > 64 KiB of dynamic LDS: raise the per-kernel limit once per device (thread-safe)
        if (lds > 48 * 1024) attr.ensure(reinterpret_cast<const void*>(&gemm_kernel<T, EPI, BM, BN, STAGES, RB, WM, WN, OCC, X3>), lds_fixed + (epi_is_rope(EPI) ? 64 * 256 : 0));
        if (a.K % (RB / (int)sizeof(T)) != 0 || (EPI == TOC3D_EPI_CONV3X3 && a.lda % (RB / (int)sizeof(T)) != 0)) {    // K-tile must divide K (conv: the channel count)
            if (RB == 128) { g_bad_variant = true; return; }
            launch_cfg<T, EPI, BM, BN, STAGES, 128, WM, WN, OCC, X3>(a, s);
            return;
        }
        const int tm = (a.M + BM - 1) / BM, tn = (a.N + BN - 1) / BN;
        int tiles = a.order == 0 ? tm * tn : 8 * ((tm + 7) / 8) * tn;           // order 1: 8 XCD bands of ceil(tm / 8) rows
        if (a.order >= 2) { const int pm = a.order == 2 ? 4 : 2, pn = 8 / pm; tiles = 8 * ((tm + pm - 1) / pm) * ((tn + pn - 1) / pn); }
        toc3d_launch((gemm_kernel<T, EPI, BM, BN, STAGES, RB, WM, WN, OCC, X3>), dim3(tiles), dim3(64 * WM * WN), lds, s, a);
    }
}

// split-K form of a tile variant (a.split = 2 .. TOC3D_SPLITK_MAX workgroups per tile; gemm_epi_splitk.hip instantiates the residual epilogues on a few tiles)
template <typename T, int EPI, int BM, int BN, int STAGES, int RB = 128, int WM = 2, int WN = 2, int OCC = 1, int X3 = 0>
void launch_cfg_sk(const GemmArgs& a, hipStream_t s) {
    constexpr int lds = STAGES * (BM + BN) * RB + (epi_ln_in(EPI) ? BM * 8 : 0);
    static Toc3dLdsAttr attr;
    if (lds > 48 * 1024) attr.ensure(reinterpret_cast<const void*>(&gemm_kernel<T, EPI, BM, BN, STAGES, RB, WM, WN, OCC, X3, 1>), lds);
    if (a.K % (RB / (int)sizeof(T)) != 0) { g_bad_variant = true; return; }       // the last slice ends at K: K must be whole K-tiles (the cuts are multiples of 128)
    const int tm = (a.M + BM - 1) / BM, tn = (a.N + BN - 1) / BN;
    toc3d_launch((gemm_kernel<T, EPI, BM, BN, STAGES, RB, WM, WN, OCC, X3, 1>), dim3(tm * tn * a.split), dim3(64 * WM * WN), lds, s, a);
}
// tile variants that have a split-K form (variant mod 1000 of toc3d_linear_fused_ws; variant / 1000 = the split): BM * BN, or 0
// (BM << 16 | BN, the shape launch_epi_sk below instantiates -- the host sizes the workspace from the SAME table: variants 9 (128x64) and 10 / 26 (64x128) have equal
// element counts and different tile grids)
constexpr int sk_tile_dims(int v) {
    return (v == 16 || v == 17 || v == 28 || v == 29 || v == 1 || v == 22) ? (128 << 16 | 128) : (v == 55 || v == 56) ? (96 << 16 | 128) : v == 19 ? (256 << 16 | 128)
           : (v == 10 || v == 26) ? (64 << 16 | 128) : v == 9 ? (128 << 16 | 64) : v == 14 ? (64 << 16 | 64) : 0;
}
constexpr int64_t sk_tile_elems(int v) { return (int64_t)(sk_tile_dims(v) >> 16) * (sk_tile_dims(v) & 0xffff); }
template <typename T, int EPI, int X3 = 0>
int launch_epi_sk(int variant, const GemmArgs& a, hipStream_t s) {
    constexpr bool B = sizeof(T) == 2;
    static_assert(sk_tile_dims(9) == (128 << 16 | 64) && sk_tile_dims(10) == (64 << 16 | 128) && sk_tile_dims(26) == (64 << 16 | 128) && sk_tile_dims(19) == (256 << 16 | 128) &&
                  sk_tile_dims(55) == (96 << 16 | 128) && sk_tile_dims(14) == (64 << 16 | 64) && sk_tile_dims(22) == (128 << 16 | 128), "sk_tile_dims must name the tiles instantiated below");
    switch (variant) {
        case 1: launch_cfg_sk<T, EPI, 128, 128, 2, 128, 2, 2, 1, X3>(a, s); break;
        case 9: launch_cfg_sk<T, EPI, 128, 64, 2, 128, 2, 2, 1, X3>(a, s); break;
        case 10: launch_cfg_sk<T, EPI, 64, 128, 2, 128, 2, 2, 1, X3>(a, s); break;
        case 14: launch_cfg_sk<T, EPI, 64, 64, 2, 128, 2, 2, 1, X3>(a, s); break;
        case 16: launch_cfg_sk<T, EPI, 128, 128, 1, 128, 2, 4, (B ? 6 : 1), X3>(a, s); break;
        case 17: launch_cfg_sk<T, EPI, 128, 128, 2, 128, 2, 4, 1, X3>(a, s); break;
        case 19: launch_cfg_sk<T, EPI, 256, 128, 1, 128, 4, 2, 1, X3>(a, s); break;
        case 22: launch_cfg_sk<T, EPI, 128, 128, 1, 256, 2, 4, 1, X3>(a, s); break;
        case 26: launch_cfg_sk<T, EPI, 64, 128, 1, 256, 2, 4, 1, X3>(a, s); break;
        case 28: launch_cfg_sk<T, EPI, 128, 128, 3, 128, 2, 4, 1, X3>(a, s); break;
        case 29: if constexpr (X3 == 0) launch_cfg_sk<T, EPI, 128, 128, 4, 128, 2, 4, 1, X3>(a, s); else return TOC3D_ERR_ARG; break;
        case 55: if constexpr (X3 == 0) launch_cfg_sk<T, EPI, 96, 128, 2, 128, 2, 4, 1, X3>(a, s); else return TOC3D_ERR_ARG; break;
        case 56: if constexpr (X3 == 0) launch_cfg_sk<T, EPI, 96, 128, 4, 128, 2, 4, 1, X3>(a, s); else return TOC3D_ERR_ARG; break;
        default: return TOC3D_ERR_ARG;
    }
    return TOC3D_OK;
}

template <int EPI, int BM, int BN, int WM, int WN, bool X3 = false>
void launch_phased(const GemmArgs& a, hipStream_t s) {
    // round 5: every linear epilogue (the folded LayerNorms' statistics in and out, the rotating q|k|v epilogue); not the conv gather (its own operand loader)
    if constexpr ((epi_is_swiglu(EPI) && (BN / WN) % 32 != 0) || (epi_stats_out(EPI) && BN % (epi_is_swiglu(EPI) ? 128 : 64) != 0) || EPI == TOC3D_EPI_CONV3X3) {
        g_bad_variant = true;
    } else {
        constexpr int lds_fixed = 2 * (BM + BN) * 128 + (epi_ln_in(EPI) ? BM * 8 : 0);              // two K-tiles of 64 bf16 (+ the row table)
        const int lds = lds_fixed + (epi_is_rope(EPI) ? (a.rope_L * 256 + 1023) / 1024 * 1024 : 0);   // + the RoPE tables (cos | sin), whole DMA instructions
        if (lds > 160 * 1024) { g_bad_variant = true; return; }
        if (X3 && !(a.a_planes && a.w_planes)) { g_bad_variant = true; return; }      // the x3 form stages planes as they lie: both operands must be planes (TOC3D_DTYPE_F32X3P)
        static Toc3dLdsAttr attr;
        attr.ensure(reinterpret_cast<const void*>(&gemm_phased_kernel<EPI, BM, BN, WM, WN, X3>), lds_fixed + (epi_is_rope(EPI) ? 64 * 256 : 0));   // (64 * 256: the largest rope_side toc3d_linear_qkv_rope accepts)
        const int tm = (a.M + BM - 1) / BM, tn = (a.N + BN - 1) / BN;
        const int tiles = a.order == 0 ? tm * tn : 8 * ((tm + 7) / 8) * tn;
        toc3d_launch((gemm_phased_kernel<EPI, BM, BN, WM, WN, X3>), dim3(tiles), dim3(512), lds, s, a);
    }
}

// tile / pipeline variants (toc3d_linear_ex `variant`); 0 = heuristic
template <typename T, int EPI>
int launch_epi(int variant, GemmArgs a, hipStream_t s) {
    if (variant >= 300) { a.order = 3; variant -= 300; }      // variant + 300: 2 row bands x 4 column quarters per XCD
    else if (variant >= 200) { a.order = 2; variant -= 200; } // variant + 200: 4 row bands x 2 column halves
    else if (variant >= 100) { a.order = 1; variant -= 100; } // variant + 100: same tile shape, per-XCD band order
    if (variant == 0) {
        // measured on MI355X (tools/gemm_sweep.py): occupancy beats ring depth on these shapes -- single-buffer tiles
        // (24-32 KiB LDS, >= 3 workgroups per CU); the narrower tile when there are few 128x128 tiles
        const int t128 = ((a.M + 127) / 128) * ((a.N + 127) / 128);
        variant = t128 < 700 ? 17 : 16;
    }
    switch (variant) {
        case 1: launch_cfg<T, EPI, 128, 128, 2>(a, s); break;
        case 8: launch_cfg<T, EPI, 128, 128, 1>(a, s); break;
        case 9: launch_cfg<T, EPI, 128, 64, 2>(a, s); break;
        case 10: launch_cfg<T, EPI, 64, 128, 2>(a, s); break;
        case 13: launch_cfg<T, EPI, 128, 64, 1>(a, s); break;
        case 14: launch_cfg<T, EPI, 64, 64, 2>(a, s); break;
        case 15: launch_cfg<T, EPI, 128, 128, 1, 128, 2, 2, 4>(a, s); break;      // v8 forced to <= 128 registers: 4 workgroups / CU
        case 16: launch_cfg<T, EPI, 128, 128, 1, 128, 2, 4, (sizeof(T) == 2 ? 6 : 1)>(a, s); break;   // 8 waves, 64x32 per wave; bf16 held to 80 registers (6 waves / SIMD = 3 workgroups / CU; the SwiGLU epilogue would take 82)
        case 17: launch_cfg<T, EPI, 128, 128, 2, 128, 2, 4, 1>(a, s); break;      // 8 waves, double buffered (64 KiB)
        case 19: launch_cfg<T, EPI, 256, 128, 1, 128, 4, 2, 1>(a, s); break;      // 8 waves, 256x128, single buffer (48 KiB)
        // big K-tiles for latency-bound launches (about one tile per CU): fewer, fatter load rounds
        case 22: launch_cfg<T, EPI, 128, 128, 1, 256, 2, 4, 1>(a, s); break;      // K-tile 128 bf16, 64 KiB
        case 24: launch_cfg<T, EPI, 64, 128, 1, 512, 2, 4, 1>(a, s); break;       // 64x128 tile, K-tile 256, 96 KiB
        case 26: launch_cfg<T, EPI, 64, 128, 1, 256, 2, 4, 1>(a, s); break;       // 64x128 tile, K-tile 128, 48 KiB
        case 27: launch_cfg<T, EPI, 64, 64, 1, 512, 2, 2, 1>(a, s); break;        // 64x64 tile, 4 waves, K-tile 256, 64 KiB
        // deep LDS rings on 8 wavefronts: more bytes continuously in flight per CU (counted vmcnt, one barrier per K-tile)
        case 28: launch_cfg<T, EPI, 128, 128, 3, 128, 2, 4, 1>(a, s); break;      // 96 KiB
        case 29: launch_cfg<T, EPI, 128, 128, 4, 128, 2, 4, 1>(a, s); break;      // 128 KiB
        case 30: if (sizeof(T) == 2) launch_cfg<bf16_t, EPI, 128, 128, 4, 64, 2, 4, 1>(a, s); else return TOC3D_ERR_ARG; break;   // K-tile 32, 64 KiB
        case 33: launch_cfg<T, EPI, 128, 64, 4, 128, 2, 4, 1>(a, s); break;       // 128x64, 4-deep, 96 KiB
        // 16 wavefronts per workgroup: 256-wide tiles (fewer L2->LDS bytes per FLOP) without giving up waves per CU
        // K-tile 32 rings on 8 wavefronts at the LDS footprint of the single-buffer tile: prefetch inside the workgroup without losing occupancy
        // N-tiles that are not powers of two (the vendor library's answer to tile-count quantisation on N = 3072 / 1024)
        case 45: launch_cfg<T, EPI, 128, 192, 2, 128, 2, 4, 1>(a, s); break;      // 128x192 double buffered, 80 KiB
        case 47: launch_cfg<T, EPI, 128, 192, 2, 128, 4, 2, 1>(a, s); break;      // 128x192 double buffered, 32x96 per wave (serves SwiGLU)
        case 49: launch_cfg<T, EPI, 192, 128, 2, 128, 2, 4, 1>(a, s); break;      // 192x128 double buffered, 96x32 per wave (serves SwiGLU), 80 KiB
        // (the rotating q|k|v epilogue is held to 128 registers -- two workgroups per CU like the other epilogues: unconstrained it took 138, ONE workgroup per CU and 83 instead of 51 us at M = 6000)
        case 52: launch_cfg<T, EPI, 192, 192, 1, 128, 2, 4, (epi_is_rope(EPI) ? 4 : 1)>(a, s); break;      // 192x192 single buffer, 96x48 per wave, 48 KiB: 512 tiles for q|k|v at M = 6000 (one per slot at two per CU)
        case 53: launch_cfg<T, EPI, 192, 192, 2, 128, 2, 4, (epi_is_rope(EPI) ? 4 : 1)>(a, s); break;      // 192x192 double buffered, 96 KiB
        // M-tiles of 96 / 160 rows (round 4): the frame's launches run 1.2-2.6 rounds of 128-row tiles on the chip's 512-768 workgroup slots and pay for a whole
        // last round; a 96- or 160-row tile changes the tile count by 4/3 or 4/5 at the same N-tile (whole statistics slots, whole (w1, w2) groups)
        case 54: launch_cfg<T, EPI, 96, 128, 1, 128, 2, 4, (sizeof(T) == 2 ? 6 : 1)>(a, s); break;    // 96x128 single buffer, 48x32 per wave, 28 KiB
        case 55: launch_cfg<T, EPI, 96, 128, 2, 128, 2, 4, 1>(a, s); break;                           // 96x128 double buffered, 56 KiB
        case 56: launch_cfg<T, EPI, 96, 128, 4, 128, 2, 4, 1>(a, s); break;                           // 96x128 4-deep ring, 112 KiB
        case 57: launch_cfg<T, EPI, 160, 128, 1, 128, 2, 4, 1>(a, s); break;                          // 160x128 single buffer, 80x32 per wave, 36 KiB
        case 58: launch_cfg<T, EPI, 160, 128, 2, 128, 2, 4, 1>(a, s); break;                          // 160x128 double buffered, 72 KiB
        case 59: launch_cfg<T, EPI, 192, 128, 3, 128, 2, 4, 1>(a, s); break;                          // 192x128 3-deep ring, 120 KiB: 32 x 8 = 256 tiles for N = 1024 at M = 6000
        case 51: launch_cfg<T, EPI, 128, 128, 1, 128, 2, 4, (sizeof(T) == 2 ? 8 : 1)>(a, s); break;   // variant 16 held to 64 registers (bf16): FOUR workgroups per CU -- the loop is bound by operand bytes in flight per CU
        // phased big tiles (bf16 only): one workgroup per CU, four phases per K-tile, the two wave groups one barrier apart
        case 60: if (sizeof(T) == 2) launch_phased<EPI, 256, 256, 2, 4>(a, s); else return TOC3D_ERR_ARG; break;   // 128x64 per wave, 128 KiB
        case 61: if (sizeof(T) == 2) launch_phased<EPI, 256, 128, 4, 2>(a, s); else return TOC3D_ERR_ARG; break;   // 64x64 per wave, 96 KiB
        case 62: if (sizeof(T) == 2) launch_phased<EPI, 128, 256, 2, 4>(a, s); else return TOC3D_ERR_ARG; break;   // 64x64 per wave, 96 KiB
        case 63: if (sizeof(T) == 2) launch_phased<EPI, 128, 128, 2, 4>(a, s); else return TOC3D_ERR_ARG; break;   // 64x32 per wave, 64 KiB: two per CU
        // register-pipelined rings (bf16 only; gemm_tile, PIPE): the next K step's fragments are read while the current one's MFMAs run -- for one workgroup per CU
#define TOC3D_PIPE(BM_, BN_, ST_, WM_, WN_)                                                                                                              \
        if constexpr (sizeof(T) == 2 && EPI != TOC3D_EPI_CONV3X3) launch_cfg<bf16_t, EPI, BM_, BN_, ST_, 128, WM_, WN_, 1, 1>(a, s); \
        else return TOC3D_ERR_ARG;                                                                                                                       \
        break
        case 64: TOC3D_PIPE(128, 128, 4, 2, 4);              // variant 29's tile and ring: 8 waves, 64x32 per wave, 128 KiB
        case 65: TOC3D_PIPE(128, 128, 3, 2, 4);              // variant 28's: 96 KiB
        case 66: TOC3D_PIPE(128, 128, 4, 2, 2);              // 4 waves, 64x64 per wave, 128 KiB
#undef TOC3D_PIPE
        default: return TOC3D_ERR_ARG;
    }
    return TOC3D_OK;
}

// bf16 x 3 / x 6 products on f32 operands (TOC3D_DTYPE_F32X3 / F32X6): a set of tile variants (same numbering as launch_epi)
template <int EPI, int X>
int launch_epi_x(int variant, GemmArgs a, hipStream_t s) {
    if (variant >= 300) { a.order = 3; variant -= 300; }
    else if (variant >= 200) { a.order = 2; variant -= 200; }
    else if (variant >= 100) { a.order = 1; variant -= 100; }
    if (variant == 0) {
        const int t128 = ((a.M + 127) / 128) * ((a.N + 127) / 128);
        variant = t128 < 700 ? 17 : 16;
    }
    // the rotating q|k|v epilogue (round 6) is held to 128 registers on the 8-wave tiles that share a CU between two workgroups (the bf16 form of variants 52 / 53 does the
    // same): unconstrained, variant 49 took 148 -- ONE workgroup per CU, 137 instead of ~75 us for q|k|v at M = 6000
    constexpr int RO = epi_is_rope(EPI) ? 4 : 1;
    switch (variant) {
        case 1: launch_cfg<float, EPI, 128, 128, 2, 128, 2, 2, 1, X>(a, s); break;
        case 8: launch_cfg<float, EPI, 128, 128, 1, 128, 2, 2, 1, X>(a, s); break;
        // round 5 (fp32x3 is the constructor default now): the tiles the bf16 tables lean on, for the x3 tables' in-place tuning -- the three-product K step is
        // longer, so tile-count quantisation weighs more
        case 9: launch_cfg<float, EPI, 128, 64, 2, 128, 2, 2, 1, X>(a, s); break;
        case 29: if constexpr (X == 3) launch_cfg<float, EPI, 128, 128, 4, 128, 2, 4, 1, X>(a, s); else return TOC3D_ERR_ARG; break;
        case 33: if constexpr (X == 3) launch_cfg<float, EPI, 128, 64, 4, 128, 2, 4, 1, X>(a, s); else return TOC3D_ERR_ARG; break;
        case 45: if constexpr (X == 3) launch_cfg<float, EPI, 128, 192, 2, 128, 2, 4, RO, X>(a, s); else return TOC3D_ERR_ARG; break;
        case 47: if constexpr (X == 3) launch_cfg<float, EPI, 128, 192, 2, 128, 4, 2, RO, X>(a, s); else return TOC3D_ERR_ARG; break;
        case 52: if constexpr (X == 3) launch_cfg<float, EPI, 192, 192, 1, 128, 2, 4, RO, X>(a, s); else return TOC3D_ERR_ARG; break;
        case 53: if constexpr (X == 3) launch_cfg<float, EPI, 192, 192, 2, 128, 2, 4, 1, X>(a, s); else return TOC3D_ERR_ARG; break;
        case 10: launch_cfg<float, EPI, 64, 128, 2, 128, 2, 2, 1, X>(a, s); break;
        case 14: launch_cfg<float, EPI, 64, 64, 2, 128, 2, 2, 1, X>(a, s); break;
        case 16: launch_cfg<float, EPI, 128, 128, 1, 128, 2, 4, 1, X>(a, s); break;
        case 17: launch_cfg<float, EPI, 128, 128, 2, 128, 2, 4, 1, X>(a, s); break;
        case 19: launch_cfg<float, EPI, 256, 128, 1, 128, 4, 2, RO, X>(a, s); break;
        case 22: launch_cfg<float, EPI, 128, 128, 1, 256, 2, 4, RO, X>(a, s); break;      // K-tile of 64 f32: half the barriers per K
        case 26: launch_cfg<float, EPI, 64, 128, 1, 256, 2, 4, 1, X>(a, s); break;
        case 28: launch_cfg<float, EPI, 128, 128, 3, 128, 2, 4, 1, X>(a, s); break;
        case 49: launch_cfg<float, EPI, 192, 128, 2, 128, 2, 4, RO, X>(a, s); break;
        // phased big tiles on planes (round 6; gemm_phased_kernel, X3): both operands must be planes (TOC3D_DTYPE_F32X3P), K a multiple of 32
        // M-tiles of 96 / 160 rows on planes (round 6: the N = 1024 residual GEMMs of the accelerated blocks make 176-392 tiles of 128 rows for 256 CUs; see the bf16 table)
        case 54: if constexpr (X == 3) launch_cfg<float, EPI, 96, 128, 1, 128, 2, 4, 1, X>(a, s); else return TOC3D_ERR_ARG; break;
        case 55: if constexpr (X == 3) launch_cfg<float, EPI, 96, 128, 2, 128, 2, 4, 1, X>(a, s); else return TOC3D_ERR_ARG; break;
        case 56: if constexpr (X == 3) launch_cfg<float, EPI, 96, 128, 4, 128, 2, 4, 1, X>(a, s); else return TOC3D_ERR_ARG; break;
        case 57: if constexpr (X == 3) launch_cfg<float, EPI, 160, 128, 1, 128, 2, 4, 1, X>(a, s); else return TOC3D_ERR_ARG; break;
        case 58: if constexpr (X == 3) launch_cfg<float, EPI, 160, 128, 2, 128, 2, 4, 1, X>(a, s); else return TOC3D_ERR_ARG; break;
        case 59: if constexpr (X == 3) launch_cfg<float, EPI, 192, 128, 3, 128, 2, 4, 1, X>(a, s); else return TOC3D_ERR_ARG; break;
        case 60: if constexpr (X == 3) launch_phased<EPI, 256, 256, 2, 4, true>(a, s); else return TOC3D_ERR_ARG; break;
        case 61: if constexpr (X == 3) launch_phased<EPI, 256, 128, 4, 2, true>(a, s); else return TOC3D_ERR_ARG; break;
        case 62: if constexpr (X == 3) launch_phased<EPI, 128, 256, 2, 4, true>(a, s); else return TOC3D_ERR_ARG; break;
        case 63: if constexpr (X == 3) launch_phased<EPI, 128, 128, 2, 4, true>(a, s); else return TOC3D_ERR_ARG; break;
        default: return TOC3D_ERR_ARG;
    }
    return TOC3D_OK;
}

}  // namespace
