// Micro-benchmark (development tool, not part of the library): which path brings GEMM operand tiles into a CU fastest on MI355X?
// Rounds 1-4 found every K loop of the GEMM family at 9-14 TB/s of L2 -> LDS fill chip-wide (35-55 GB/s per CU) whatever the tile -- this
// probe separates the paths.  One workgroup = one 128x128 tile position (m0, n0) walking K in 64-element steps like the GEMM does:
//   mode 0: 16-byte global_load_lds (LDS-DMA), DEPTH K-tiles in flight (counted vmcnt, raw barrier)
//   mode 1: 16-byte global_load to VGPRs, values discarded (what the L1 -> register path delivers)
//   mode 2: global_load to VGPRs + ds_write_b128 into the LDS tile (register staging)
//   mode 3: A rows by LDS-DMA, W rows by VGPR + ds_write (both paths at once)
//   mode 4: A rows by LDS-DMA, W rows to VGPRs and discarded
// Operands are random bf16 (DVFS: zero-filled operands clock higher, cdna_hip_programming.md rule 25).
// Reports operand GB/s per CU and the equivalent TFLOP/s of 128x128x64 steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int DEPTH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const char* __restrict__ A, const char* __restrict__ B, int64_t ld, int nk, int tiles_n, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128, RB = 128, NTHR = 64 * WAVES, PER = BM * 8 / NTHR;    // 16-byte pieces per thread per operand per K-tile
    constexpr int STAGE = 2 * BM * RB;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // XCD-aware: consecutive tiles of one row panel on one XCD (block b runs on XCD b % 8)
    const int nwg = gridDim.x, per = nwg / 8;
    const int tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BM;
    u32x4 acc = u32x4{0, 0, 0, 0};
    auto src_of = [&](const char* g0, int row0, int kt, int t) {
        const int cidx = t * NTHR + wave * 64 + lane, r = cidx >> 3, p = cidx & 7;
        return g0 + (int64_t)(row0 + r) * ld + kt * RB + ((p ^ (r & 7)) << 4);
    };
    auto dma = [&](const char* g0, int row0, int kt, char* dst) {
#pragma unroll
        for (int t = 0; t < PER; ++t)
            __builtin_amdgcn_global_load_lds((gptr_t)src_of(g0, row0, kt, t), (lptr_t)(dst + (t * NTHR + wave * 64) * 16), 16, 0, 0);
    };
    u32x4 ra[DEPTH][PER], rb[DEPTH][PER];
    auto vload = [&](const char* g0, int row0, int kt, u32x4 (&r)[PER]) {
#pragma unroll
        for (int t = 0; t < PER; ++t) r[t] = *reinterpret_cast<const u32x4*>(src_of(g0, row0, kt, t));
    };
    auto vsink = [&](u32x4 (&r)[PER], char* dst, bool write) {
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            if (write) *reinterpret_cast<u32x4*>(dst + (t * NTHR + wave * 64 + lane) * 16) = r[t];
            else acc ^= r[t];
        }
    };
    constexpr bool A_DMA = MODE == 0 || MODE == 3 || MODE == 4, B_DMA = MODE == 0;
    constexpr bool WRITE = MODE == 2 || MODE == 3;
    auto request = [&](int kt, auto S) {
        constexpr int s = decltype(S)::value;
        char* slot = smem + s * STAGE;
        if constexpr (A_DMA) dma(A, m0, kt, slot); else vload(A, m0, kt, ra[s]);
        if constexpr (B_DMA) dma(B, n0, kt, slot + BM * RB); else vload(B, n0, kt, rb[s]);
    };
    auto consume = [&](auto S) {
        constexpr int s = decltype(S)::value;
        char* slot = smem + s * STAGE;
        if constexpr (!A_DMA) vsink(ra[s], slot, WRITE);
        if constexpr (!B_DMA) vsink(rb[s], slot + BM * RB, WRITE);
    };
    // software ring of DEPTH stages, unrolled so every register array index is static
    constexpr int LOADS = 2 * PER;
#define STEP(S)                                                                                  \
    {                                                                                            \
        using I = std::integral_constant<int, S>;                                                \
        if (kt + S < nk) {                                                                       \
            if (nk - 1 - (kt + S) >= DEPTH - 1) wait_vmcnt<(DEPTH - 1) * LOADS>();               \
            else wait_vmcnt<0>();                                                                \
            consume(I());                                                                        \
            __builtin_amdgcn_s_barrier();                                                        \
            if (kt + S + DEPTH < nk) request(kt + S + DEPTH, I());                               \
        }                                                                                        \
    }
    {
        using I0 = std::integral_constant<int, 0>;
        request(0, I0());                                  // nk >= DEPTH for every shape of main()
        if constexpr (DEPTH > 1) { using I1 = std::integral_constant<int, 1>; request(1, I1()); }
        if constexpr (DEPTH > 2) { using I2 = std::integral_constant<int, 2>; request(2, I2()); }
        if constexpr (DEPTH > 3) { using I3 = std::integral_constant<int, 3>; request(3, I3()); }
    }
    for (int kt = 0; kt < nk; kt += DEPTH) {
        STEP(0)
        if constexpr (DEPTH > 1) STEP(1)
        if constexpr (DEPTH > 2) STEP(2)
        if constexpr (DEPTH > 3) STEP(3)
    }
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = acc[0];
}

static float* g_flush;
template <int MODE, int DEPTH, int WAVES>
void run(const char* name, const char* A, const char* B, int M, int N, int K, int wg_per_cu, unsigned* sink) {
    const int tm = M / 128, tn = N / 128, nk = K * 2 / 128;
    size_t lds = DEPTH * 32768;
    const size_t want = 163840 / wg_per_cu / 1024 * 1024;
    if (want < lds) { printf("%-28s depth=%d: %d WG/CU does not fit\n", name, DEPTH, wg_per_cu); return; }
    if (wg_per_cu < 5 && 163840 / (wg_per_cu + 1) >= lds) lds = 163840 / (wg_per_cu + 1) + 1024;   // pad so that exactly wg_per_cu fit
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, DEPTH, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, DEPTH, WAVES>), dim3(tm * tn), dim3(64 * WAVES), lds, 0, A, B, (int64_t)K * 2, nk, tn, sink);
    hipEventRecord(e0);
    const int R = 6;
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL((k<MODE, DEPTH, WAVES>), dim3(tm * tn), dim3(64 * WAVES), lds, 0, A, B, (int64_t)K * 2, nk, tn, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= R;
    const double bytes = (double)tm * tn * nk * 32768.0, flops = 2.0 * M * N * K;
    const int cus = tm * tn < 256 ? tm * tn : 256;
    printf("%-28s depth=%d waves=%d WG/CU<=%d lds=%6zu  %8.1f us  %6.1f GB/s/CU (%5.2f TB/s)  equiv %5.0f TF (128x128 tiles), %5.0f TF (if 256x256)\n", name, DEPTH, WAVES,
           wg_per_cu, lds, ms * 1e3, bytes / (ms * 1e-3) / cus / 1e9, bytes / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12, 2 * flops / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

template <int WAVES>
void sweep(const char* A, const char* B, int M, int N, int K, unsigned* sink) {
    printf("=== M=%d N=%d K=%d (%d tiles of 128x128, %d K-steps), %d waves per WG\n", M, N, K, (M / 128) * (N / 128), K / 64, WAVES);
    for (int wpc : {1, 2, 3, 4}) {
        run<0, 1, WAVES>("LDS-DMA", A, B, M, N, K, wpc, sink);
        run<0, 2, WAVES>("LDS-DMA", A, B, M, N, K, wpc, sink);
        if (wpc <= 1) run<0, 4, WAVES>("LDS-DMA", A, B, M, N, K, wpc, sink);
        run<1, 1, WAVES>("VGPR discard", A, B, M, N, K, wpc, sink);
        run<1, 2, WAVES>("VGPR discard", A, B, M, N, K, wpc, sink);
        if (wpc <= 2) run<1, 4, WAVES>("VGPR discard", A, B, M, N, K, wpc, sink);
        run<2, 2, WAVES>("VGPR + ds_write", A, B, M, N, K, wpc, sink);
        run<3, 2, WAVES>("A DMA | W VGPR+ds_write", A, B, M, N, K, wpc, sink);
        run<4, 2, WAVES>("A DMA | W VGPR discard", A, B, M, N, K, wpc, sink);
    }
}

int main() {
    const int Mmax = 6144, Nmax = 3072, Kmax = 4096;
    char *A, *B; unsigned* sink;
    hipMalloc(&A, (size_t)Mmax * Kmax * 2); hipMalloc(&B, (size_t)Nmax * Kmax * 2); hipMalloc(&sink, 4);
    std::vector<unsigned short> h((size_t)Mmax * Kmax);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((s >> 9) & 0x3ff) + ((s >> 31) << 15)); }   // random bf16 in +-[0.0078, 0.0156)
    hipMemcpy(A, h.data(), (size_t)Mmax * Kmax * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)Nmax * Kmax * 2, hipMemcpyHostToDevice);
    sweep<8>(A, B, 6144, 3072, 1024, sink);      // the frame's q|k|v shape: 1152 tiles, 16 K-steps
    sweep<8>(A, B, 6144, 2048, 4096, sink);      // 768 tiles, 64 K-steps: steady state
    sweep<4>(A, B, 6144, 2048, 4096, sink);
    sweep<8>(A, B, 2048, 2048, 4096, sink);      // 256 tiles: one per CU
    return 0;
}
