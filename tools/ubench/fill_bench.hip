// Micro-benchmark (development tool, not part of the library): what limits the GEMM's operand staging on MI355X?
//   mode 0: global_load_lds fill only (A/B tile pattern of gemm.hip, swizzled source), no LDS reads, no MFMA
//   mode 1: fill + the ds_read_b128 fragment reads of an 8-wave 128x128 tile (no MFMA)
//   mode 2: fill + reads + MFMAs (the real inner loop, results discarded into a dummy store)
//   mode 3: reads + MFMAs only (LDS filled once)
// Reports GB/s per CU of operand bytes and the equivalent TFLOP/s for 128x128x64 steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const char* __restrict__ A, const char* __restrict__ B, int64_t ld, int nk, int tiles_n, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128, BN = 128, RB = 128, NTHR = 64 * WAVES;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, g = lane >> 4;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    constexpr int WN = WAVES == 8 ? 4 : 2, TM = 64, TN = BN / WN, MT = TM / 16, NT = TN / 16;
    const int wm = wave / WN, wn = wave % WN;
    f32x4 acc[MT][NT];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    auto stage = [&](const char* g0, int row0, int k0, char* dst) {
#pragma unroll
        for (int t = 0; t < BM * 8 / NTHR; ++t) {
            const int cidx = t * NTHR + wave * 64 + lane, r = cidx >> 3, p = cidx & 7;
            const char* src = g0 + (int64_t)(row0 + r) * ld + k0 + ((p ^ (r & 7)) << 4);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + (t * NTHR + wave * 64) * 16), 16, 0, 0);
        }
    };
    if (MODE == 3) { stage(A, m0, 0, smem); stage(B, n0, 0, smem + BM * RB); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    for (int kt = 0; kt < nk; ++kt) {
        if (MODE != 3) {
            stage(A, m0, kt * RB, smem);
            stage(B, n0, kt * RB, smem + BM * RB);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (MODE >= 1) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 fa[MT], fb[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) { const int r = wm * TM + i * 16 + r16; fa[i] = *reinterpret_cast<const bf16x8*>(smem + r * RB + (((s * 4 + g) ^ (r & 7)) << 4)); }
#pragma unroll
                for (int j = 0; j < NT; ++j) { const int r = wn * TN + j * 16 + r16; fb[j] = *reinterpret_cast<const bf16x8*>(smem + BM * RB + r * RB + (((s * 4 + g) ^ (r & 7)) << 4)); }
                if (MODE >= 2) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i][0][0] += (float)fa[i][0];
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[0][j][1] += (float)fb[j][0];
                }
            }
        }
        if (MODE != 3) __builtin_amdgcn_s_barrier();
    }
    float t = 0;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123.456f) sink[0] = t;
}

template <int MODE, int WAVES>
void run(const char* name, const char* A, const char* B, int M, int N, int K, int pad_lds, float* sink) {
    const int tm = M / 128, tn = N / 128, nk = K * 2 / 128;
    const size_t lds = 32768 + pad_lds;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, WAVES>), dim3(tm * tn), dim3(64 * WAVES), lds, 0, A, B, (int64_t)K * 2, nk, tn, sink);
    hipEventRecord(e0);
    const int R = 10;
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL((k<MODE, WAVES>), dim3(tm * tn), dim3(64 * WAVES), lds, 0, A, B, (int64_t)K * 2, nk, tn, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= R;
    const double bytes = (double)tm * tn * nk * 32768.0, flops = 2.0 * M * N * K;
    printf("%-34s waves=%d lds=%6zu  %8.1f us  fill %6.1f GB/s/CU (%5.2f TB/s)  equiv %6.0f TF\n", name, WAVES, lds, ms * 1e3, bytes / (ms * 1e-3) / 256 / 1e9,
           bytes / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12);
}

int main() {
    const int M = 6016, N = 3072, K = 1024;
    char *A, *B; float* sink;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&sink, 4);
    hipMemset(A, 0, (size_t)M * K * 2); hipMemset(B, 0, (size_t)N * K * 2);
    for (int pad : {0, 8192, 22000, 50000, 130000}) {
        printf("--- LDS per WG %d B (%d WG/CU by LDS)\n", 32768 + pad, 163840 / (32768 + pad));
        run<0, 8>("fill only", A, B, M, N, K, pad, sink);
        run<1, 8>("fill + ds_read", A, B, M, N, K, pad, sink);
        run<2, 8>("fill + ds_read + mfma", A, B, M, N, K, pad, sink);
        run<3, 8>("ds_read + mfma (no fill)", A, B, M, N, K, pad, sink);
        run<2, 4>("fill + ds_read + mfma", A, B, M, N, K, pad, sink);
        run<3, 4>("ds_read + mfma (no fill)", A, B, M, N, K, pad, sink);
    }
    return 0;
}
