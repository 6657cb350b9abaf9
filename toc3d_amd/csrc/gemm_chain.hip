// GEMM chains: the dependent linear layers of one transformer-block half in ONE persistent launch (bf16 path).
//
// Reference ops served: attn.proj + residual -> norm2 -> mlp.w1 | mlp.w2 -> ffn_ln -> mlp.w3 + residual
// (eva_vit.py:44-51,115,262-263 / toc3d_eva_vit.py:366-386), i.e. the launches toc3d_linear_fused issues one by one.
//
// Why: as separate launches every GEMM of the frame runs in lock step -- all workgroups stream operands and multiply, then all of them
// store at once with the matrix cores idle, then the chip drains and the next launch ramps (per-workgroup timelines, tools/ubench/gemm_timeline.hip),
// and the N = 1024 GEMMs have fewer tiles than the chip has workgroup slots.  Here a grid of resident workgroups pulls tiles from queues
// that hold the tiles of ALL the GEMMs of the chain in dependency order; a tile of a later GEMM starts as soon as the row panels it reads
// are complete, so the epilogue stores of some tiles overlap the K loops of others and the small GEMMs fill the tails of the big ones.
//
// Scheduling.  The host cuts the M-tiles into *bands* and writes one tile list per band (the `schedule` argument: any order in which a
// tile's producers come first).  A band is claimed by ONE XCD (the hardware's XCC_ID of the claiming workgroup) and only workgroups running
// on that XCD pull from it, so a row panel is produced and consumed through one L2: the hand-off of a panel from GEMM to GEMM needs no
// L2 write-back, only the consumer's L1 invalidate.  An XCD claims another band when its bands have no tiles left to hand out, so the
// result does not depend on how the dispatcher spread the grid (an XCD that received no workgroup simply never claims a band); correctness
// never depends on blockIdx -> XCD placement.  full_release = 1 adds the agent-scope release in front of every publish (the form that
// would also be correct if a band's tiles ran on several XCDs).
//
// Hand-off (cdna_hip_programming.md Guideline 16, counter form): producer tile: every wave drains its stores (s_waitcnt vmcnt(0)) ->
// workgroup barrier -> one lane adds 1 to done[op][M-tile] (relaxed, agent scope).  Consumer tile: one lane polls the counters of the
// producer M-tiles that cover its rows (relaxed loads, s_sleep between polls, bounded) -> ONE agent-scope acquire -> barrier -> operand
// loads.  Every tile is computed by one workgroup with the K order of gemm_tile, so the results are bit-identical to the separate launches.
//
// State words are zero before the first launch and are zeroed again by the last workgroup to leave (no memset node in the recorded frame).
#include "gemm_kernels.h"

namespace {

typedef __attribute__((address_space(1))) unsigned gu32;
#define TOC3D_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

template <int EPI_, int BM_, int BN_, int STAGES_, int RB_, int WM_, int WN_>
struct Tile {
    static constexpr int EPI = EPI_, BM = BM_, BN = BN_, STAGES = STAGES_, RB = RB_, WM = WM_, WN = WN_;
    static constexpr int THREADS = 64 * WM_ * WN_;
    static constexpr int LDS = STAGES_ * (BM_ + BN_) * RB_ + (epi_ln_in(EPI_) ? BM_ * 8 : 0);
};
struct NoTile {
    static constexpr int EPI = -1, BM = 128, BN = 128, STAGES = 1, RB = 128, WM = 2, WN = 4, THREADS = 512, LDS = 0;
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }

enum { CHAIN_ERR_SPIN = 1, CHAIN_ERR_SCHED = 2 };

// OCC: the kernel's register cap, as gemm_kernel passes it -- a capped tile forms its (mean, rstd) table before anything else of the tile is live.
// (Round 4: with the table's batch of statistics loads behind the first operand request -- the uncapped form -- family 1 / tiling 0 came back wrong in the
// lanes hipcc keeps spilled SGPRs in: 46 SGPR + 77 VGPR spills under the 80-register cap, tools/gpu/chain_dbg.py.  No product kernel spills an SGPR.)
template <typename C, int OCC>
TOC3D_DEV void chain_tile(const GemmArgs& a, int mt, int nt, char* smem) {
    if constexpr (C::EPI >= 0) gemm_tile<bf16_t, C::EPI, C::BM, C::BN, C::STAGES, C::RB, C::WM, C::WN, 0, OCC>(a, mt * C::BM, nt * C::BN, smem);
}

template <typename C0, typename C1, typename C2, int OCC>
__global__ __launch_bounds__(512, OCC) void gemm_chain_kernel(ChainArgs c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(C0::THREADS == 512 && C1::THREADS == 512 && C2::THREADS == 512, "chain tiles are 8-wavefront tiles");
    constexpr int LDS_TILES = cmax(C0::LDS, cmax(C1::LDS, C2::LDS));
    volatile int* mbox = reinterpret_cast<volatile int*>(smem + LDS_TILES);       // the one LDS array of the kernel (a second __shared__ object de-pipelines the K loops)
    const int tid = threadIdx.x;
    gu32* state = (gu32*)c.state;
    gu32* owner = state + 8;
    gu32* head = state + 8 + TOC3D_CHAIN_MAX_BANDS;
    gu32* done = state + 8 + 2 * TOC3D_CHAIN_MAX_BANDS;
    const unsigned me = __builtin_amdgcn_s_getreg((3 << 11) | 20) + 1;            // HW_REG_XCC_ID[3:0] + 1: the XCD this workgroup really runs on
    int band = -1, scan = 0;                                                       // lane 0's cursor: the band it pulls from, the first band worth looking at

    for (;;) {
        unsigned long long tr[5];
        if (c.trace && tid == 0) tr[0] = __builtin_amdgcn_s_memrealtime();
        // ---- one lane fetches the next tile: from the XCD's current band, else from the next band this XCD owns or can still claim ----
        if (tid == 0) {
            int entry = -1;
            for (;;) {
                if (band >= 0) {
                    const unsigned idx = __hip_atomic_fetch_add(head + band, 1u, TOC3D_RLX_AGENT);
                    const unsigned cnt = (unsigned)c.sched[2 * band + 1];
                    if (idx < cnt) { entry = c.sched[c.sched[2 * band] + idx]; break; }
                    scan = band + 1;
                    band = -1;
                }
                int found = -1;
                for (int b = scan; b < c.n_bands; ++b) {
                    unsigned o = __hip_atomic_load(owner + b, TOC3D_RLX_AGENT);
                    if (o == 0) {
                        unsigned expect = 0;
                        o = __hip_atomic_compare_exchange_strong(owner + b, &expect, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? me : expect;
                    }
                    if (o == me && __hip_atomic_load(head + b, TOC3D_RLX_AGENT) < (unsigned)c.sched[2 * b + 1]) { found = b; break; }
                }
                if (found < 0) break;
                band = scan = found;
            }
            *mbox = entry;
        }
        lds_barrier();
        const int entry = *mbox;
        if (entry < 0) break;
        const int op = (entry >> 28) & 7, mt = (entry >> 16) & 0xfff, nt = entry & 0xffff;
        if (c.trace && tid == 0) tr[1] = __builtin_amdgcn_s_memrealtime();
        // ---- wait for the row panels this tile reads ----
        const int dep = op == 0 ? c.op[0].dep : (op == 1 ? c.op[1].dep : c.op[2].dep);
        if (dep >= 0) {
            if (tid == 0) {
                const ChainOp& o = op == 1 ? c.op[1] : (op == 2 ? c.op[2] : c.op[0]);
                const int bm = op == 0 ? C0::BM : (op == 1 ? C1::BM : C2::BM);
                const int r_lo = mt * bm, r_hi = (mt * bm + bm - 1 < o.a.M - 1 ? mt * bm + bm - 1 : o.a.M - 1);
                unsigned spins = 0;
                for (int p = r_lo / o.dep_bm; p <= r_hi / o.dep_bm; ++p) {
                    while (__hip_atomic_load(done + dep * TOC3D_CHAIN_MAX_MT + p, TOC3D_RLX_AGENT) < (unsigned)o.dep_need) {
                        if (++spins > c.max_polls) { __hip_atomic_store(state + 1, (unsigned)CHAIN_ERR_SPIN, TOC3D_RLX_AGENT); break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");               // drops this CU's stale L1 lines; the loads below are plain
            }
            lds_barrier();
        }
        if (c.trace && tid == 0) tr[2] = __builtin_amdgcn_s_memrealtime();
        if (op == 0) chain_tile<C0, OCC>(c.op[0].a, mt, nt, smem);
        else if (op == 1) chain_tile<C1, OCC>(c.op[1].a, mt, nt, smem);
        else if (op == 2) chain_tile<C2, OCC>(c.op[2].a, mt, nt, smem);
        else if (tid == 0) __hip_atomic_store(state + 1, (unsigned)CHAIN_ERR_SCHED, TOC3D_RLX_AGENT);
        if (c.trace && tid == 0) tr[3] = __builtin_amdgcn_s_memrealtime();
        // ---- publish the tile: its stores are acknowledged by the L2 before the counter moves ----
        const int publish = op == 0 ? c.op[0].publish : (op == 1 ? c.op[1].publish : c.op[2].publish);
        if (publish) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // every storing wave
            lds_barrier();
            if (tid == 0) {
                if (c.full_release) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the wait hipcc may drop behind buffer_wbl2 (Guideline 16, pitfall 12)
                }
                __hip_atomic_fetch_add(done + op * TOC3D_CHAIN_MAX_MT + mt, 1u, TOC3D_RLX_AGENT);
            }
        }
        if (c.trace && tid == 0) {
            tr[4] = __builtin_amdgcn_s_memrealtime();
            const unsigned long long slot = __hip_atomic_fetch_add(c.trace, 1ull, TOC3D_RLX_AGENT);
            if (slot < (unsigned long long)c.trace_cap) {
                unsigned long long* t = c.trace + 1 + slot * 8;
                t[0] = ((unsigned long long)me << 56) | ((unsigned long long)blockIdx.x << 32) | (unsigned)entry;
                for (int q = 0; q < 5; ++q) t[1 + q] = tr[q];
                t[6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
            }
        }
    }
    // ---- the last workgroup to leave re-arms the state for the next launch (nobody polls any more) ----
    if (tid == 0) {
        const unsigned left = __hip_atomic_fetch_add(state, 1u, TOC3D_RLX_AGENT);
        if (left == gridDim.x - 1) {
            for (int w = 2; w < TOC3D_CHAIN_STATE_WORDS; ++w) __hip_atomic_store(state + w, 0u, TOC3D_RLX_AGENT);
            __hip_atomic_store(state, 0u, TOC3D_RLX_AGENT);
        }
    }
}

template <typename C> void tile_info(int* info) { info[0] = C::EPI; info[1] = C::BM; info[2] = C::BN; info[3] = C::THREADS; }

template <typename C0, typename C1, typename C2, int OCC>
int chain_cfg(ChainArgs* c, int grid, hipStream_t s, int* info) {
    constexpr int n_ops = C2::EPI >= 0 ? 3 : 2;
    if (info) { tile_info<C0>(info); tile_info<C1>(info + 4); tile_info<C2>(info + 8); return n_ops; }
    if (c->n_ops != n_ops) return TOC3D_ERR_ARG;
    const int epis[3] = {C0::EPI, C1::EPI, C2::EPI}, bms[3] = {C0::BM, C1::BM, C2::BM}, bns[3] = {C0::BN, C1::BN, C2::BN};
    for (int i = 0; i < n_ops; ++i) {
        ChainOp& o = c->op[i];
        if (o.dep >= i) return TOC3D_ERR_ARG;
        if ((o.a.M + bms[i] - 1) / bms[i] > TOC3D_CHAIN_MAX_MT) return TOC3D_ERR_ARG;
        if (o.a.K % 64 != 0 || o.a.M != c->op[0].a.M) return TOC3D_ERR_ARG;          // every op works on the same rows
        (void)epis;
        if (o.dep >= 0) { o.dep_bm = bms[o.dep]; o.dep_need = (c->op[o.dep].a.N + bns[o.dep] - 1) / bns[o.dep]; }
    }
    constexpr int lds = cmax(C0::LDS, cmax(C1::LDS, C2::LDS)) + 64;
    static Toc3dLdsAttr attr;
    if (lds > 48 * 1024) attr.ensure(reinterpret_cast<const void*>(&gemm_chain_kernel<C0, C1, C2, OCC>), lds);
    toc3d_launch((gemm_chain_kernel<C0, C1, C2, OCC>), dim3(grid), dim3(512), lds, s, *c);
    return TOC3D_OK;
}

// chain configurations: family (the epilogues) x tiling.  config = 10 * family + tiling
//   family 0: attn.proj (RESIDUAL_STATS) -> w1|w2 (SWIGLU_STATS_LN) -> w3 (RESIDUAL_LN)      norm2 and ffn_ln folded
//   family 1: w1|w2 (SWIGLU_STATS) -> w3 (RESIDUAL_LN)                                       ffn_ln folded, norm2 as its own launch in front
template <int E0, int E1, int E2>
int chain_family(int tiling, ChainArgs* c, int grid, hipStream_t s, int* info) {
    using T2no = NoTile;
    switch (tiling) {
        // 128x128 single-buffer tiles (32 KB), three workgroups per CU
        case 0: if constexpr (E2 >= 0) return chain_cfg<Tile<E0, 128, 128, 1, 128, 2, 4>, Tile<E1, 128, 128, 1, 128, 2, 4>, Tile<E2, 128, 128, 1, 128, 2, 4>, 6>(c, grid, s, info);
                else return chain_cfg<Tile<E0, 128, 128, 1, 128, 2, 4>, Tile<E1, 128, 128, 1, 128, 2, 4>, T2no, 6>(c, grid, s, info);
        // the N = 1024 GEMMs on 64x128 tiles (twice the tiles: they have fewer tiles than the chip has slots)
        case 1: if constexpr (E2 >= 0) return chain_cfg<Tile<E0, 64, 128, 1, 128, 2, 4>, Tile<E1, 128, 128, 1, 128, 2, 4>, Tile<E2, 64, 128, 1, 128, 2, 4>, 6>(c, grid, s, info);
                else return chain_cfg<Tile<E0, 128, 128, 1, 128, 2, 4>, Tile<E1, 64, 128, 1, 128, 2, 4>, T2no, 6>(c, grid, s, info);
        // double-buffered 128x128 tiles (64 KB), two workgroups per CU
        case 2: if constexpr (E2 >= 0) return chain_cfg<Tile<E0, 128, 128, 2, 128, 2, 4>, Tile<E1, 128, 128, 2, 128, 2, 4>, Tile<E2, 128, 128, 2, 128, 2, 4>, 4>(c, grid, s, info);
                else return chain_cfg<Tile<E0, 128, 128, 2, 128, 2, 4>, Tile<E1, 128, 128, 2, 128, 2, 4>, T2no, 4>(c, grid, s, info);
        default: return TOC3D_ERR_ARG;
    }
}

int chain_dispatch(int config, ChainArgs* c, int grid, hipStream_t s, int* info) {
    const int family = config / 10, tiling = config % 10;
    switch (family) {
        case 0: return chain_family<TOC3D_EPI_RESIDUAL_STATS, TOC3D_EPI_SWIGLU_STATS_LN, TOC3D_EPI_RESIDUAL_LN>(tiling, c, grid, s, info);
        case 1: return chain_family<TOC3D_EPI_SWIGLU_STATS, TOC3D_EPI_RESIDUAL_LN, -1>(tiling, c, grid, s, info);
        default: return TOC3D_ERR_ARG;
    }
}

}  // namespace

static std::atomic<unsigned long long*> g_chain_trace{nullptr};
static std::atomic<int> g_chain_trace_cap{0};
void toc3d_gemm_chain_set_trace(void* buf, int entries) { g_chain_trace_cap.store(entries); g_chain_trace.store(reinterpret_cast<unsigned long long*>(buf)); }

int toc3d_gemm_chain_launch(int config, ChainArgs& c, int grid, hipStream_t s) {
    c.trace = g_chain_trace.load();
    c.trace_cap = g_chain_trace_cap.load();
    return chain_dispatch(config, &c, grid, s, nullptr);
}
int toc3d_gemm_chain_info(int config, int* info) {
    const int rc = chain_dispatch(config, nullptr, 0, nullptr, info);
    return rc > 0 ? rc : -1;
}
