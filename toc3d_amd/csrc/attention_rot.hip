// Windowed multi-head self-attention on PRE-ROTATED q / k (bf16, gfx950): the RoPE of q and k and the scale of q -- head_dim^-0.5 * log2(e): the
// softmax here is exp2-based (round 5) -- were applied by the q|k|v projection's epilogue (gemm_kernels.h, EPI_QKV_ROPE), so this kernel does no
// arithmetic on its operands before the MFMAs.
//
// Reference: backbones/eva_vit.py:101-113 (dense windows), backbones/toc3d_eva_vit.py:499-512 (kept tokens + representative token,
// RoPE rows gathered by slot index, backbones/eva_utils.py:396-403).
//
// One workgroup = one (window, head): the window's K and V rows of that head go HBM/L2 -> LDS by 16-byte global_load_lds straight
// from the q|k|v buffer (no VGPR round trip, no conversion, no transposing 2-byte stores: the staging of the r02 kernels was ~1300 VALU
// instructions per wave and 42 % LDS bank conflicts, profiles/r02_attention_pmc_summary.txt), every load of the workgroup in flight at
// once behind ONE dependent index read.  Both images are row-major [key][64 dims] with the 16-byte chunks of a row XOR-permuted on the
// SOURCE address (the DMA image is lane-linear, cdna_hip_programming.md rule 21):
//   K: chunk c of row r at position c ^ (r & 7)             -> conflict-free ds_read_b128 of the S^T = K.Q^T A fragments;
//   V: chunk c of row r at position c ^ (((r >> 1) & 3) * 2) -> conflict-free ds_read_b64_tr_b16: the hardware transposing read hands
//      lane (d, g) the values V[4 keys of group g][dim d], i.e. the A fragment of O^T = V^T.P^T, from the row-major image.
// Scores are computed transposed (a lane owns ONE query, its keys spread over the 4 lane groups), P stays in registers and is the B
// operand of O^T = V^T.P^T, whose accumulator gives a lane 4 consecutive output dims of its own query: the softmax statistics never
// leave the lane (one butterfly over the lane groups) and the output leaves in 8-byte stores.
//
// Virtual kept-pad keys of accelerated blocks (rows[j] < 0): their q|k|v is the projection of LN(0) = beta, a per-block constant, but
// the RoPE of its k depends on the window slot -- the host packs pad_rot [window slots, 3C] with the same GEMM epilogue, and the key
// reads row slots[j] of it.  Dense blocks: the npad zero-pad keys are folded analytically (attention.hip header).
#include "capi.h"
#include "common.h"

// Development instrumentation (tools/ubench/attn_timeline.py builds a private copy with -DTOC3D_ATTN_TRACE; the library never defines it):
// per-workgroup stamps of the 100 MHz real-time counter at entry / indices back / operands landed / compute done / stores acknowledged.
#ifdef TOC3D_ATTN_TRACE
__device__ unsigned long long* toc3d_attn_trace_buf;     // [workgroup][8]
extern "C" int toc3d_attn_trace_set(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(toc3d_attn_trace_buf), &p, sizeof(p)); }
#define ATTN_TRACE(slot)                                                                                                        \
    do {                                                                                                                        \
        asm volatile("" ::: "memory");                                                                                          \
        if (threadIdx.x == 0 && toc3d_attn_trace_buf)                                                                           \
            toc3d_attn_trace_buf[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
        asm volatile("" ::: "memory");                                                                                          \
    } while (0)
#else
#define ATTN_TRACE(slot) do {} while (0)
#endif

// Cache policy of the K / V DMA loads: 2 = nt.  Every K / V byte of the q|k|v buffer is read by exactly one workgroup, once; streamed past the L2s' LRU it leaves the
// prefetched weights and the GEMM operands alone: +0.3-0.4 % frames/s, eight alternations on two boxes, all in favour (profiles/r05_nt_stores.txt; the same hint on the row
// kernels' read-once rows LOSES 2 %, on the GEMM's W operand 19.5 %).  Bit-identical.
#ifndef TOC3D_ATTN_KV_AUX
#define TOC3D_ATTN_KV_AUX 2
#endif

namespace {

constexpr int HD = 64;
constexpr float NEG_BIG = -1.0e30f;
constexpr float RESCALE_THR = 8.0f;              // log2 units: the softmax reference point of a query moves only when a chunk's maximum exceeds it by more than this
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct AttnRotArgs {
    const bf16_t* qkv; int64_t ldqkv;
    bf16_t* out; int64_t ldo;
    const int32_t* rows; const int32_t* slots; const int32_t* count; const int32_t* count_k; const int32_t* npad;
    const bf16_t* pad_rot;
    int64_t stride;
    int C;
    const float* v_bias;
    // weight prefetch riding on this launch: every wavefront pulls pf_instr KB of these buffers through the caches (see prefetch_weights)
    const char* pf_ptr[4]; int64_t pf_n16[4]; int pf_instr;
};

// The attention kernels leave HBM idle, and the GEMMs that follow them start on weights that were last touched a frame ago (0.6 GB of bf16
// weights cycle through a 256 MB Infinity Cache).  Round 2 streamed the next GEMMs' weights through extra "rider" workgroups in front of the
// attention grid; they held CU slots and registers and delayed the attention workgroups behind them by 3-6 us per launch
// (development timeline of that version; profiles/r03_attn_timeline.txt is the shipped kernel).  Here every wavefront of the attention grid itself requests a few KB of those weights by LDS-DMA into a
// 1 KB dump area right after its operands have landed: no registers, no extra workgroups, the requests drain while the wave computes.  Wave w
// of W takes the 1 KB pieces w, w + W, ... (neighbouring waves read neighbouring KB).
TOC3D_DEV void prefetch_weights(const AttnRotArgs& a, char* dump, int64_t wave_id, int64_t nwaves, int lane) {
    for (int j = 0; j < a.pf_instr; ++j) {
        int64_t c = ((int64_t)j * nwaves + wave_id) * 64 + lane;      // 16-byte piece of the concatenated buffers
        const char* src = nullptr;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            if (!src && c < a.pf_n16[sg]) src = a.pf_ptr[sg] + c * 16;
            c -= a.pf_n16[sg];
        }
        if (!src) src = a.pf_ptr[0];                                  // past the end: any valid line
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dump, 16, 0, 0);
    }
}

// (g4_max: the maximum over the four lane groups by v_permlane16_swap / v_permlane32_swap -- common.h; it sits on every 32-key chunk's dependent chain)

TOC3D_DEV int v_swz(int r) { return ((r >> 1) & 3) << 1; }

// One instantiation serves every window size up to 416 keys.  Rounds 3-4 computed the scores TWICE (a first pass for the exact row maximum, a second one
// for P = exp(S - max)) to stay at 72 registers = 7 waves per SIMD with ONE query tile per wavefront; the first version had kept every score of a query
// tile in registers (94-176 VGPRs, 3-4 waves per SIMD) and was latency-bound at 9-25 us per workgroup.  Round 5: ONE pass -- an online softmax whose
// reference point only moves when a chunk's maximum exceeds it by more than RESCALE_THR (see the loop) -- at the same 72 registers: a third of the
// MFMAs and K-fragment reads and ~a quarter of the VALU instructions less per query tile (profiles/r05_attention.txt).
constexpr int MAXPC = 8;                         // DMA pieces (8 keys x 128 B) per wave and operand: the host launches >= ceil(keys / 64) waves

__global__ __launch_bounds__(1024, 7) void attn_rot_kernel(AttnRotArgs a) {      // 72 VGPRs: 7 waves per SIMD, 28 per CU
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NW = blockDim.x >> 6;
    const int head = blockIdx.x, win = (int)blockIdx.y;
    ATTN_TRACE(0);
    const int n = a.count[win];
    if (n == 0) return;
    const int nkeys = a.count_k ? a.count_k[win] : n;
    const int NK32 = ((nkeys + 31) >> 5) << 5;   // keys padded to the 32-wide P.V step
    char* Ks = smem;                             // [NK32][128 B]
    char* Vs = smem + NK32 * 128;                // [NK32][128 B]
    const int32_t* rows = a.rows + (int64_t)win * a.stride;
    const int32_t* slots = a.slots + (int64_t)win * a.stride;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r16 = lane & 15, g = lane >> 4;

    // ---- 1. the one dependent index round trip: rows of this lane's DMA pieces and of its (at most two) query tiles ----
    const int nmt = (n + 15) >> 4;
    int krow[MAXPC];                             // >= 0: row of the q|k|v buffer; < 0: virtual pad key, -1 - (window slot)
#pragma unroll
    for (int i = 0; i < MAXPC; ++i) {
        const int piece = i * NW + wave;         // wave-uniform; one piece = 8 keys x 8 chunks = 1 KB per operand
        krow[i] = 0;
        if (piece * 8 < NK32) {
            int key = piece * 8 + (lane >> 3);
            key = key < nkeys ? key : 0;         // rows past the list alias key 0: finite values, their scores are masked
            int row = rows[key];
            if (a.pad_rot) { const int sl = slots[key]; row = row >= 0 ? row : -1 - sl; }
            krow[i] = row;
        }
    }
    int qrow[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int qi = (wave + NW * u) * 16 + r16;
        qrow[u] = rows[qi < n ? qi : 0];         // the first `count` entries are real rows
    }
#ifdef TOC3D_ATTN_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATTN_TRACE(1);
#endif
    // ---- 2. everything in flight at once: Q fragments to registers, K and V rows to LDS by DMA ----
    Frag<bf16_t> qf[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
            if (wave + NW * u < nmt) qf[u][s2] = read_frag(a.qkv + (int64_t)qrow[u] * a.ldqkv + head * HD + s2 * 32 + g * 8);
#pragma unroll
    for (int i = 0; i < MAXPC; ++i) {
        const int piece = i * NW + wave;
        if (piece * 8 < NK32) {
            const int r = piece * 8 + (lane >> 3), p = lane & 7;
            const bf16_t* src = krow[i] >= 0 ? a.qkv + (int64_t)krow[i] * a.ldqkv : a.pad_rot + (int64_t)(-1 - krow[i]) * a.ldqkv;
            const char* kb = reinterpret_cast<const char*>(src + a.C + head * HD);
            const char* vb = reinterpret_cast<const char*>(src + 2 * a.C + head * HD);
            __builtin_amdgcn_global_load_lds((gptr_t)(kb + ((p ^ (r & 7)) << 4)), (lptr_t)(Ks + piece * 1024), 16, 0, TOC3D_ATTN_KV_AUX);
            __builtin_amdgcn_global_load_lds((gptr_t)(vb + ((p ^ v_swz(r)) << 4)), (lptr_t)(Vs + piece * 1024), 16, 0, TOC3D_ATTN_KV_AUX);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ATTN_TRACE(2);
    if (a.pf_instr > 0)                          // dump area: 1 KB behind the V image
        prefetch_weights(a, Vs + NK32 * 128, ((int64_t)win * gridDim.x + head) * NW + wave, (int64_t)gridDim.y * gridDim.x * NW, lane);

    const int np = a.npad ? a.npad[win] : 0;
    // K fragment of 16-key tile t, 32-dim half s2: row t*16 + r16, chunk s2*4 + g (permuted by row & 7 = r16 & 7)
    const char* kf0 = Ks + r16 * 128 + ((g ^ (r16 & 7)) << 4);
    const char* kf1 = Ks + r16 * 128 + (((4 + g) ^ (r16 & 7)) << 4);
    // V^T fragment: this lane addresses 4 dims of key row g*4 + (r16 >> 2) (+ 16 for the second half) of each 32-key chunk; the
    // chunk permutation depends on that row only, not on the chunk
    const int vr = g * 4 + (r16 >> 2);
    const char* vbase = Vs + vr * 128 + (r16 & 1) * 8;
    const int vsw = v_swz(vr), vch = (r16 & 3) >> 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int mt = wave + NW * u;
        if (mt >= nmt) break;
        // ---- ONE pass over the keys, 32 at a time (round 5; rounds 3-4 computed every score twice: a first pass for the exact row maximum).  Online softmax with a
        // DEFERRED maximum (cdna_hip_programming.md T13): the reference point m of a query only moves when a chunk's maximum exceeds it by more than RESCALE_THR
        // (then sum and O^T are rescaled once); otherwise P = exp2(S - m) <= 2^RESCALE_THR stays far inside the f32 / bf16 range.  Softmax is invariant to m, so
        // the result is the exact softmax up to rounding.  q arrives scaled by head_dim^-0.5 * log2(e) (the q|k|v epilogue), so exp2 needs no multiply.
        // lane holds S[q = r16][keys of its lane group]; key slot (g, j < 4) = key c*32 + g*4 + j, (g, j >= 4) = key c*32 + 16 + g*4 + (j - 4) for both operands ----
        f32x4 o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        float sum = 0.f;
        float mx = np > 0 ? 0.f : NEG_BIG;       // the analytic zero-pad keys score 0
        const int nfull = nkeys >> 5;            // chunks without a masked key
        auto chunk = [&](const int c, const bool tail) {
            f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = f32x4{0.f, 0.f, 0.f, 0.f};
            Frag<bf16_t> k00, k01, k10, k11;
            k00.v = *reinterpret_cast<const bf16x8*>(kf0 + c * 4096);
            k01.v = *reinterpret_cast<const bf16x8*>(kf1 + c * 4096);
            k10.v = *reinterpret_cast<const bf16x8*>(kf0 + c * 4096 + 2048);
            k11.v = *reinterpret_cast<const bf16x8*>(kf1 + c * 4096 + 2048);
            mma_step(s0, k00, qf[u][0]);
            mma_step(s1, k10, qf[u][0]);
            mma_step(s0, k01, qf[u][1]);
            mma_step(s1, k11, qf[u][1]);
            if (tail) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s0[r] = c * 32 + g * 4 + r < nkeys ? s0[r] : NEG_BIG;
                    s1[r] = c * 32 + 16 + g * 4 + r < nkeys ? s1[r] : NEG_BIG;
                }
            }
            const float cmax = g4_max(fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3]))));
            if (__builtin_amdgcn_ballot_w64(cmax > mx + RESCALE_THR) != 0ull) {          // wave-uniform: some query's maximum moved (always in the first chunk)
                const float mnew = fmaxf(mx, cmax);                                     // per query; a query whose maximum did not move rescales by exactly 1
                const float sc = __builtin_amdgcn_exp2f(mx - mnew);
                sum *= sc;
#pragma unroll
                for (int d = 0; d < 4; ++d) o[d] *= sc;
                mx = mnew;
            }
            float pv[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { pv[r] = __builtin_amdgcn_exp2f(s0[r] - mx); pv[4 + r] = __builtin_amdgcn_exp2f(s1[r] - mx); }
            const Frag<bf16_t> pf = make_frag(pv, bf16_t());
            sum += ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const char* p0 = vbase + c * 4096 + (((d * 2 + vch) ^ vsw) << 4);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lptr_t)p0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lptr_t)(p0 + 2048));
                Frag<bf16_t> vf;
                vf.v = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                mma_step(o[d], vf, pf);          // rows = head dims d*16 + .., columns = queries
            }
        };
        for (int c = 0; c < nfull; ++c) chunk(c, false);
        if ((nfull << 5) < NK32) chunk(nfull, true);                                    // the one chunk that reaches past the key list
        sum = g4_sum(sum);
        float padw = 0.f;
        if (np > 0) { padw = (float)np * __builtin_amdgcn_exp2f(-mx); sum += padw; }
        const float inv = 1.f / sum;
        const int qi = mt * 16 + r16;
        if (qi < n) {
            bf16_t* dst = a.out + (int64_t)qrow[u] * a.ldo + head * HD + g * 4;
#pragma unroll
            for (int d = 0; d < 4; ++d) {        // o[d][r] = O[q = r16][dim d*16 + g*4 + r]
                bf16_t o4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = o[d][r];
                    if (np > 0) v += padw * a.v_bias[head * HD + d * 16 + g * 4 + r];
                    o4[r] = to_act<bf16_t>(v * inv);
                }
                store4(dst + d * 16, o4);
            }
        }
    }
    ATTN_TRACE(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the prefetch DMAs target this workgroup's LDS: they must have landed before it is released
    ATTN_TRACE(4);
}

// Wavefronts per workgroup: one 16-query tile per wave where the chip can hold the whole grid at once, two otherwise.  Always a multiple of
// four: a workgroup's waves are dealt to the four SIMDs in turn, and the kernel's 72 VGPRs admit 7 waves per SIMD -- with 9 waves per
// workgroup the third workgroup of a CU found no SIMD-balanced home and the launch ran in 1.5 rounds (development timeline of that version).
void launch_rot(const AttnRotArgs& a, int64_t max_count, int64_t num_heads, int64_t nwin, hipStream_t s) {
    const size_t lds = (size_t)((a.stride + 31) / 32 * 32) * 256 + 1024;      // K image, V image, 1 KB prefetch dump
    static Toc3dLdsAttr attr;
    attr.ensure(reinterpret_cast<const void*>(&attn_rot_kernel), 112 * 1024);
    const int nqt = (int)((max_count + 15) / 16);
    const int64_t wgs = nwin * num_heads;
    const int by_lds = (int)(160 * 1024 / (lds > 0 ? lds : 1));
    int per_cu = (int)((wgs + 255) / 256);                   // workgroups per CU if the whole grid is to be resident
    per_cu = per_cu < 1 ? 1 : (per_cu > by_lds ? by_lds : per_cu);
    int quads = 7 / per_cu;                                   // waves per SIMD and workgroup
    quads = quads < 1 ? 1 : (quads > 4 ? 4 : quads);
    const int want = (nqt + 3) / 4;                           // one query tile per wave
    quads = quads > want ? want : quads;
    int least = ((nqt + 1) / 2 + 3) / 4;                      // two query tiles per wave at most ...
    const int by_keys = (int)((a.stride + 255) / 256);        // ... and at most MAXPC = 8 DMA pieces of 8 keys per wave
    least = least < by_keys ? by_keys : least;
    quads = quads < least ? least : quads;
    AttnRotArgs b = a;
    if (b.pf_instr > 0) {                                     // KB per wavefront so that the grid covers the buffers once (at most 8)
        int64_t kb = 0;
        for (int i = 0; i < 4; ++i) kb += (b.pf_n16[i] + 63) / 64;
        const int64_t waves = wgs * 4 * quads;
        const int64_t per = (kb + waves - 1) / waves;
        b.pf_instr = (int)(per > 8 ? 8 : per);
    }
    toc3d_launch(attn_rot_kernel, dim3((unsigned)num_heads, (unsigned)nwin), dim3(256 * quads), lds, s, b);
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The same attention on (hi, lo) bf16 PLANES (round 6; precision "fp32x3", TOC3D_DTYPE_F32X3P): the q|k|v buffer was written by the x3 projection's
// rotating epilogue as planes (common.h: element c of a row in the 128-byte group c / 32, hi at byte 2 (c % 32), lo at 64 + 2 (c % 32)), i.e. a head's
// 64 dims are 256 bytes = 16 chunks [hi 0..31 | lo 0..31 | hi 32..63 | lo 32..63].  Both contractions are bf16 x 3 products like the GEMMs of that precision
// (a . b = lo.hi + hi.lo + hi.hi, small terms first), softmax statistics and accumulation in f32, P split in registers once per 32 keys.
//   K image [key][256 B], chunk c of key r at position c ^ (r & 15): the ds_read_b128 lane groups (16 rows x one chunk) hit 16 distinct 16-byte slots;
//   V image [key][256 B], chunk c of key r at position c ^ ((r & 7) << 1): the 8 keys of a ds_read_b64_tr_b16 lane group (2 x 16 lanes, 32 bytes per key) hit
//   8 distinct 32-byte slots of the 256-byte bank row.
// A key costs 512 bytes of LDS, so windows of more than `tile_keys` keys are walked in SUPER-TILES of tile_keys keys (MULTI): stage, barrier, every
// query tile of the wave over the staged keys (the online softmax state of both tiles stays in registers), barrier, next -- the 400-key global windows of
// the dense blocks in 128-key tiles at two workgroups per CU.  Windows that fit are staged whole behind one dependent index read, like the bf16 kernel.
struct AttnRotX3Args {
    const float* qkv; int64_t ldqkv;             // planes; leading dimension in f32 elements
    float* out; int64_t ldo;                     // planes (the projection GEMM's A operand)
    const int32_t* rows; const int32_t* slots; const int32_t* count; const int32_t* count_k; const int32_t* npad;
    const float* pad_rot;
    int64_t stride;
    int C;
    const float* v_bias;
    int tile_keys;                               // keys per super-tile (multiple of 32)
};

constexpr int X3_MAXPC = 8;                      // DMA pieces (4 keys x 256 B) per wave, operand and super-tile: the host launches >= ceil(tile_keys / 32) waves

TOC3D_DEV void x3_mma16(f32x4& acc, const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl) {      // small terms first, like the GEMM
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}

template <bool MULTI>
__global__ __launch_bounds__(1024, 4) void attn_rot_x3_kernel(AttnRotX3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NW = blockDim.x >> 6;
    const int head = blockIdx.x, win = (int)blockIdx.y;
    const int n = a.count[win];
    if (n == 0) return;
    const int nkeys = a.count_k ? a.count_k[win] : n;
    const int NK32 = ((nkeys + 31) >> 5) << 5;
    const int TK = MULTI ? a.tile_keys : NK32;   // keys staged at a time
    char* Ks = smem;                             // [TK][256 B]
    char* Vs = smem + (MULTI ? a.tile_keys : NK32) * 256;
    const int32_t* rows = a.rows + (int64_t)win * a.stride;
    const int32_t* slots = a.slots + (int64_t)win * a.stride;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r16 = lane & 15, g = lane >> 4;
    const int nmt = (n + 15) >> 4;
    // query tiles of this wave: two at most in the whole-window kernel (mt = wave, wave + NW); ONE in the super-tile kernel, whose grid carries a third dimension over
    // groups of NW query tiles instead (every group stages the window's keys again, L2-hot: the online-softmax state of two tiles beside the staging registers does
    // not fit the 128 registers of a 16-wave workgroup)
    constexpr int NU = MULTI ? 1 : 2;
    constexpr int NPC = MULTI ? 2 : X3_MAXPC;    // DMA pieces per wave, operand and staging round (MULTI: 16 waves x 2 pieces x 4 keys = the 128-key super-tile)
    const int mt0 = MULTI ? (int)blockIdx.z * NW + wave : wave;
    if (MULTI && (int)blockIdx.z * NW >= nmt) return;            // (workgroup-uniform)

    // index round trip of the first (only) super-tile + the query rows
    int krow[NPC];
    auto key_rows = [&](const int k0, const int tk) {
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int piece = i * NW + wave;     // wave-uniform; one piece = 4 keys x 16 chunks = 1 KB per operand
            krow[i] = 0;
            if (piece * 4 < tk) {
                int key = k0 + piece * 4 + (lane >> 4);
                key = key < nkeys ? key : 0;     // rows past the list alias key 0: finite values, their scores are masked
                int row = rows[key];
                if (a.pad_rot) { const int sl = slots[key]; row = row >= 0 ? row : -1 - sl; }
                krow[i] = row;
            }
        }
    };
    auto stage = [&](const int tk) {
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int piece = i * NW + wave;
            if (piece * 4 < tk) {
                const int r = piece * 4 + (lane >> 4), p = lane & 15;
                const float* src = krow[i] >= 0 ? a.qkv + (int64_t)krow[i] * a.ldqkv : a.pad_rot + (int64_t)(-1 - krow[i]) * a.ldqkv;
                const char* kb = reinterpret_cast<const char*>(src + a.C + head * HD);
                const char* vb = reinterpret_cast<const char*>(src + 2 * a.C + head * HD);
                __builtin_amdgcn_global_load_lds((gptr_t)(kb + ((p ^ (r & 15)) << 4)), (lptr_t)(Ks + piece * 1024), 16, 0, TOC3D_ATTN_KV_AUX);
                __builtin_amdgcn_global_load_lds((gptr_t)(vb + ((p ^ ((r & 7) << 1)) << 4)), (lptr_t)(Vs + piece * 1024), 16, 0, TOC3D_ATTN_KV_AUX);
            }
        }
    };
    const int tk0 = NK32 < TK ? NK32 : TK;
    key_rows(0, tk0);
    int qrow[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int qi = (mt0 + NW * u) * 16 + r16;
        qrow[u] = rows[qi < n ? qi : 0];         // the first `count` entries are real rows
    }
    // Q fragments (hi, lo) of the wave's query tiles, in flight together with the K / V DMA
    bf16x8 qh[NU][2], ql[NU][2];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            qh[u][s2] = bf16x8{}; ql[u][s2] = bf16x8{};
            if (mt0 + NW * u < nmt) {
                const char* qb = reinterpret_cast<const char*>(a.qkv + (int64_t)qrow[u] * a.ldqkv + head * HD) + s2 * 128 + g * 16;
                qh[u][s2] = *reinterpret_cast<const bf16x8*>(qb);
                ql[u][s2] = *reinterpret_cast<const bf16x8*>(qb + 64);
            }
        }
    stage(tk0);

    const int np = a.npad ? a.npad[win] : 0;
    // K fragment of the 16-key tile at key row kr (multiple of 16), 32-dim half s2: row kr + r16, hi chunk s2*8 + g, lo chunk s2*8 + 4 + g, permuted by r16
    const char* kf = Ks + r16 * 256;
    int kofs[2][2];                              // [s2][hi | lo] byte offset inside the row
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) { kofs[s2][0] = ((s2 * 8 + g) ^ r16) << 4; kofs[s2][1] = ((s2 * 8 + 4 + g) ^ r16) << 4; }
    // V^T fragment: this lane addresses 4 dims of key row g*4 + (r16 >> 2) (+ 16 for the second half) of each 32-key chunk
    const int vr = g * 4 + (r16 >> 2);
    const char* vbase = Vs + vr * 256 + (r16 & 1) * 8;
    const int vsw = (vr & 7) << 1, vch = (r16 & 3) >> 1;

    f32x4 o[NU][4];
    float sum[NU], mx[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
        for (int d = 0; d < 4; ++d) o[u][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        sum[u] = 0.f;
        mx[u] = np > 0 ? 0.f : NEG_BIG;          // the analytic zero-pad keys score 0
    }
    // one 32-key chunk of the staged keys for query tile U: local chunk c (LDS rows c*32 ..), global key offset kg = k0 + c*32
    auto chunk = [&](auto U, const int c, const int kg, const bool tail) {
        constexpr int u = decltype(U)::value;
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 k0h = *reinterpret_cast<const bf16x8*>(kf + c * 8192 + kofs[s2][0]);
            const bf16x8 k0l = *reinterpret_cast<const bf16x8*>(kf + c * 8192 + kofs[s2][1]);
            const bf16x8 k1h = *reinterpret_cast<const bf16x8*>(kf + c * 8192 + 4096 + kofs[s2][0]);
            const bf16x8 k1l = *reinterpret_cast<const bf16x8*>(kf + c * 8192 + 4096 + kofs[s2][1]);
            x3_mma16(s0, k0h, k0l, qh[u][s2], ql[u][s2]);
            x3_mma16(s1, k1h, k1l, qh[u][s2], ql[u][s2]);
        }
        if (tail) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0[r] = kg + g * 4 + r < nkeys ? s0[r] : NEG_BIG;
                s1[r] = kg + 16 + g * 4 + r < nkeys ? s1[r] : NEG_BIG;
            }
        }
        const float cmax = g4_max(fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3]))));
        if (__builtin_amdgcn_ballot_w64(cmax > mx[u] + RESCALE_THR) != 0ull) {
            const float mnew = fmaxf(mx[u], cmax);
            const float sc = __builtin_amdgcn_exp2f(mx[u] - mnew);
            sum[u] *= sc;
#pragma unroll
            for (int d = 0; d < 4; ++d) o[u][d] *= sc;
            mx[u] = mnew;
        }
        float pv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { pv[r] = __builtin_amdgcn_exp2f(s0[r] - mx[u]); pv[4 + r] = __builtin_amdgcn_exp2f(s1[r] - mx[u]); }
        bf16x8 ph, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bf16_t h = (bf16_t)pv[e];
            ph[e] = h;
            pl[e] = (bf16_t)(pv[e] - (float)h);
        }
        sum[u] += ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            // dims d*16 + (r16 & 3)*4 ..: 32-dim group d >> 1, chunk (d & 1)*2 + vch inside it (hi), + 4 (lo)
            const int ch = (d >> 1) * 8 + (d & 1) * 2 + vch;
            const char* ph0 = vbase + c * 8192 + ((ch ^ vsw) << 4);
            const char* pl0 = vbase + c * 8192 + (((ch + 4) ^ vsw) << 4);
            const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lptr_t)ph0);
            const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lptr_t)(ph0 + 4096));
            const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lptr_t)pl0);
            const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lptr_t)(pl0 + 4096));
            const bf16x8 vh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
            const bf16x8 vl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
            x3_mma16(o[u][d], vh, vl, ph, pl);   // rows = head dims d*16 + .., columns = queries
        }
    };
    auto tile_keys_of = [&](auto U, const int k0, const int tk) {
        constexpr int u = decltype(U)::value;
        if (mt0 + NW * u >= nmt) return;
        const int nfull = (nkeys - k0 < tk ? (nkeys - k0 > 0 ? nkeys - k0 : 0) : tk) >> 5;      // chunks of this super-tile without a masked key
        for (int c = 0; c < nfull; ++c) chunk(U, c, k0 + c * 32, false);
        if ((nfull << 5) < tk) chunk(U, nfull, k0 + nfull * 32, true);                          // the one chunk that reaches past the key list
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (!MULTI) {
        tile_keys_of(U0(), 0, NK32);
        tile_keys_of(U1(), 0, NK32);
    } else {
        for (int k0 = 0; k0 < NK32; k0 += TK) {
            const int tk = NK32 - k0 < TK ? NK32 - k0 : TK;
            if (k0 > 0) {
                key_rows(k0, tk);
                stage(tk);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            tile_keys_of(U0(), k0, tk);
            if (k0 + TK < NK32) __syncthreads();                  // every wave is done reading: the images may be overwritten
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int mt = mt0 + NW * u;
        if (mt >= nmt) break;
        float s = g4_sum(sum[u]);
        float padw = 0.f;
        if (np > 0) { padw = (float)np * __builtin_amdgcn_exp2f(-mx[u]); s += padw; }
        const float inv = 1.f / s;
        const int qi = mt * 16 + r16;
        // o[d][r] = O[q = r16][dim d*16 + g*4 + r].  (hi, lo) planes of the row, 16 bytes per store: the dim tiles d, d + 1 of one query are exchanged between neighbouring lane
        // groups (v_permlane16_swap: an even group ends with 8 consecutive dims of tile d, an odd one with 8 of tile d + 1) -- 8 stores per lane instead of 16
        char* drow = reinterpret_cast<char*>(a.out + (int64_t)qrow[u] * a.ldo + head * HD);
#pragma unroll
        for (int d = 0; d < 4; d += 2) {
            typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            unsigned w[2][2][2];                 // [tile d | d + 1][hi | lo][2 words]
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bf16x4_t hi, lo;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = o[u][d + t][r];
                    if (np > 0) v += padw * a.v_bias[head * HD + (d + t) * 16 + g * 4 + r];
                    v *= inv;
                    const bf16_t h = (bf16_t)v;
                    hi[r] = h;
                    lo[r] = (bf16_t)(v - (float)h);
                }
                const unsigned long long uh = __builtin_bit_cast(unsigned long long, hi), ul = __builtin_bit_cast(unsigned long long, lo);
                w[t][0][0] = (unsigned)uh; w[t][0][1] = (unsigned)(uh >> 32);
                w[t][1][0] = (unsigned)ul; w[t][1][1] = (unsigned)(ul >> 32);
            }
            // dims (d + t)*16 + g*4 .. of the 32-dim group d >> 1: byte (t*16 + g*4) * 2 of the hi plane; even groups store tile d at their own dims, odd groups tile d + 1
            // four dims to the left (their left neighbour's)
            char* p = drow + (d >> 1) * 128 + ((g & 1) ? (16 + (g - 1) * 4) * 2 : g * 8);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const auto rx = __builtin_amdgcn_permlane16_swap(w[0][pl][0], w[1][pl][0], false, false);
                const auto ry = __builtin_amdgcn_permlane16_swap(w[0][pl][1], w[1][pl][1], false, false);
                if (qi < n) *reinterpret_cast<u32x4*>(p + pl * 64) = u32x4{rx[0], ry[0], rx[1], ry[1]};
            }
        }
    }
}

void launch_rot_x3(AttnRotX3Args a, int64_t max_count, int64_t num_heads, int64_t nwin, hipStream_t s) {
    const int64_t nk32 = (a.stride + 31) / 32 * 32;
    const bool multi = nk32 * 512 > 144 * 1024;     // the window's K and V planes (512 B per key) do not fit one workgroup's LDS: walk the keys in super-tiles
    a.tile_keys = multi ? 128 : (int)nk32;          // (the super-tile kernel's staging is sized for 128 keys on 16 waves)
    const size_t lds = (size_t)a.tile_keys * 512;
    static Toc3dLdsAttr attr_s, attr_m;
    if (multi) attr_m.ensure(reinterpret_cast<const void*>(&attn_rot_x3_kernel<true>), 144 * 1024);
    else attr_s.ensure(reinterpret_cast<const void*>(&attn_rot_x3_kernel<false>), 144 * 1024);
    const int nqt = (int)((max_count + 15) / 16);
    if (multi) {
        // one query tile per wave, 16-wave workgroups, the grid's third dimension over groups of 16 query tiles
        toc3d_launch(attn_rot_x3_kernel<true>, dim3((unsigned)num_heads, (unsigned)nwin, (unsigned)((nqt + 15) / 16)), dim3(1024), lds, s, a);
        return;
    }
    const int64_t wgs = nwin * num_heads;
    const int by_lds = (int)(160 * 1024 / lds);
    int per_cu = (int)((wgs + 255) / 256);                   // workgroups per CU if the whole grid is to be resident
    per_cu = per_cu < 1 ? 1 : (per_cu > by_lds ? by_lds : per_cu);
    int quads = 4 / per_cu;                                   // waves per SIMD and workgroup (the kernel is held to 128 registers: 4 waves per SIMD)
    quads = quads < 1 ? 1 : quads;
    const int want = (nqt + 3) / 4;                           // one query tile per wave
    quads = quads > want ? want : quads;
    int least = ((nqt + 1) / 2 + 3) / 4;                      // two query tiles per wave at most ...
    const int by_keys = (int)((a.tile_keys + 127) / 128);     // ... and at most X3_MAXPC = 8 DMA pieces of 4 keys per wave
    least = least < by_keys ? by_keys : least;
    quads = quads < least ? least : quads;
    toc3d_launch(attn_rot_x3_kernel<false>, dim3((unsigned)num_heads, (unsigned)nwin), dim3(256 * quads), lds, s, a);
}

}  // namespace

extern "C" {

int toc3d_window_attention_rot(int dtype, const void* qkv, int64_t ldqkv, void* out, int64_t ldo, const int32_t* rows, const int32_t* slots,
                               const int32_t* count, const int32_t* count_k, const int32_t* npad, const void* pad_rot, int64_t stride,
                               int64_t nwin, int64_t max_count, int64_t num_heads, const float* v_bias,
                               int64_t n_prefetch, const void* const* prefetch_ptrs, const int64_t* prefetch_bytes, int64_t prefetch_workgroups,
                               toc3d_stream_t stream) {
    TOC3D_REQUIRE(dtype == TOC3D_BF16 || dtype == TOC3D_F32X3P, "toc3d_window_attention_rot: bf16, or (hi, lo) planes of the bf16 x 3 precision (TOC3D_DTYPE_F32X3P); exact f32 is toc3d_window_attention");
    if (dtype == TOC3D_F32X3P) {
        // q|k|v and out in planes (written by toc3d_linear_qkv_rope with TOC3D_DTYPE_F32X3P / F32X3WO; out = the projection GEMM's A operand); leading dimensions in f32 elements
        TOC3D_REQUIRE(qkv && out && rows && count, "toc3d_window_attention_rot: null buffer");
        TOC3D_REQUIRE(!npad || v_bias, "toc3d_window_attention_rot: npad given without v_bias");
        TOC3D_REQUIRE(!count_k || (pad_rot && slots), "toc3d_window_attention_rot: count_k given without pad_rot / slots");
        TOC3D_REQUIRE(num_heads > 0 && nwin >= 0 && max_count >= 0 && stride >= max_count, "toc3d_window_attention_rot: bad dims");
        TOC3D_REQUIRE(stride <= 1024 && max_count <= 512, "toc3d_window_attention_rot: windows of up to 1024 keys / 512 queries");
        const int64_t Cx = num_heads * HD;
        TOC3D_REQUIRE(ldqkv >= 3 * Cx && ldo >= Cx && ldqkv % 32 == 0 && ldo % 32 == 0, "toc3d_window_attention_rot: rows of planes are whole 32-element groups (ldqkv, ldo multiples of 32)");
        TOC3D_REQUIRE(((uintptr_t)qkv % 128) == 0 && ((uintptr_t)out % 128) == 0 && (!pad_rot || ((uintptr_t)pad_rot % 128) == 0), "toc3d_window_attention_rot: planes start on 128-byte boundaries");
        TOC3D_REQUIRE(num_heads <= 65535 && nwin <= 65535, "toc3d_window_attention_rot: grid too large");
        TOC3D_REQUIRE(n_prefetch == 0 || prefetch_workgroups == 0, "toc3d_window_attention_rot: the planes form carries no weight prefetch");
        if (nwin == 0 || max_count == 0) return TOC3D_OK;
        AttnRotX3Args x{(const float*)qkv, ldqkv, (float*)out, ldo, rows, slots ? slots : rows, count, count_k, npad, (const float*)pad_rot, stride, (int)Cx, v_bias, 0};
        launch_rot_x3(x, max_count, num_heads, nwin, as_stream(stream));
        TOC3D_LAUNCH_CHECK("toc3d_window_attention_rot");
        return TOC3D_OK;
    }
    TOC3D_REQUIRE(qkv && out && rows && count, "toc3d_window_attention_rot: null buffer");
    TOC3D_REQUIRE(!npad || v_bias, "toc3d_window_attention_rot: npad given without v_bias");
    TOC3D_REQUIRE(!count_k || (pad_rot && slots), "toc3d_window_attention_rot: count_k given without pad_rot / slots");
    TOC3D_REQUIRE(num_heads > 0 && nwin >= 0 && max_count >= 0 && stride >= max_count, "toc3d_window_attention_rot: bad dims");
    TOC3D_REQUIRE(stride <= 416 && max_count <= 512, "toc3d_window_attention_rot: windows of up to 416 keys / 512 queries (use toc3d_window_attention)");
    const int64_t C = num_heads * HD;
    TOC3D_REQUIRE(ldqkv >= 3 * C && ldo >= C, "toc3d_window_attention_rot: leading dims too small for head_dim 64");
    TOC3D_REQUIRE((ldqkv * 2) % 16 == 0 && ((uintptr_t)qkv % 16) == 0 && (!pad_rot || ((uintptr_t)pad_rot % 16) == 0), "toc3d_window_attention_rot: q|k|v rows must be 16-byte aligned");
    TOC3D_REQUIRE(ldo % 4 == 0 && ((uintptr_t)out % 8) == 0, "toc3d_window_attention_rot: out must be 8-byte aligned with ldo a multiple of 4");
    TOC3D_REQUIRE(num_heads <= 65535 && nwin <= 65535, "toc3d_window_attention_rot: grid too large");
    if (nwin == 0 || max_count == 0) return TOC3D_OK;
    TOC3D_REQUIRE(n_prefetch >= 0 && n_prefetch <= 4 && (n_prefetch == 0 || (prefetch_ptrs && prefetch_bytes)), "toc3d_window_attention_rot: at most 4 prefetch buffers (host arrays)");
    AttnRotArgs a{(const bf16_t*)qkv, ldqkv, (bf16_t*)out, ldo, rows, slots ? slots : rows, count, count_k, npad, (const bf16_t*)pad_rot, stride, (int)C, v_bias,
                  {nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}, 0};
    int64_t pf_total = 0;
    for (int64_t i = 0; i < n_prefetch; ++i) {
        TOC3D_REQUIRE(prefetch_bytes[i] >= 0 && ((uintptr_t)prefetch_ptrs[i] % 16) == 0, "toc3d_window_attention_rot: prefetch buffers must be 16-byte aligned");
        a.pf_ptr[i] = (const char*)prefetch_ptrs[i];
        a.pf_n16[i] = prefetch_bytes[i] / 16;
        pf_total += a.pf_n16[i];
    }
    a.pf_instr = (pf_total > 0 && prefetch_workgroups != 0) ? 1 : 0;        // launch_rot sizes it
    launch_rot(a, max_count, num_heads, nwin, as_stream(stream));
    TOC3D_LAUNCH_CHECK("toc3d_window_attention_rot");
    return TOC3D_OK;
}

}  // extern "C"
