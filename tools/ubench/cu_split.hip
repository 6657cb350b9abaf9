// Development experiment (not part of the library): do two independent half-frame kernel chains on DISJOINT halves of the CUs beat one
// full-frame chain on all CUs?
//
// Why it could: the per-workgroup timelines (profiles/r03_gemm_timeline_*.txt) show every launch of the frame running in lock step -- all
// workgroups stream operands and multiply for 20-35 us with the memory side nearly idle, then all of them store at once for 5-12 us with the
// matrix cores idle, then the chip drains and the next launch ramps.  The frame's six views are independent (two view groups of three),
// so two chains could fill each other's idle phases -- if they do not fight for the same CUs.  hipExtStreamCreateWithCUMask pins a stream to
// a CU set; this tool measures   (a) one stream, full-size GEMM chain   (b) two unmasked streams, half-size chains
// (c) two streams masked to complementary CU halves   and prints which CUs a masked kernel really ran on.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTOC3D_GEMM_TRACE -Itoc3d_amd/csrc tools/ubench/cu_split.hip -o tools/ubench/bin/cu_split
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <map>
#include <set>
#include <vector>

#include "gemm_kernels.h"

thread_local bool g_bad_variant = false;
thread_local Toc3dPlan* toc3d_tls_recording = nullptr;
void toc3d_plan_record(Toc3dPlan*, const void*, dim3, dim3, size_t, hipStream_t, const void* const*, const size_t*, const size_t*, int) {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Bufs { bf16_t *a, *qkv, *att, *hid; float* x; };

static Bufs alloc_bufs(size_t M) {
    Bufs b;
    CK(hipMalloc(&b.a, M * 1024 * 2)); CK(hipMalloc(&b.qkv, M * 3072 * 2)); CK(hipMalloc(&b.att, M * 1024 * 2)); CK(hipMalloc(&b.hid, M * 5504 * 2));
    CK(hipMalloc(&b.x, M * 1024 * 4));
    CK(hipMemset(b.a, 0x3c, M * 1024 * 2)); CK(hipMemset(b.att, 0x3c, M * 1024 * 2)); CK(hipMemset(b.hid, 0x3c, M * 5504 * 2)); CK(hipMemset(b.x, 0, M * 1024 * 4));
    return b;
}

struct Weights { bf16_t *wqkv, *wproj, *w12, *w3; float* bias; };

static void gemm(int epi, int variant, const void* A, int lda, const void* W, int K, void* out, int ldo, int M, int N, float* res, const float* bias, hipStream_t s) {
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = W; a.ldw = K; a.bias = bias; a.out = out; a.ldo = ldo; a.res = res; a.ldr = N; a.M = M; a.N = N; a.K = K; a.vec = 1;
    g_bad_variant = false;
    int rc = epi == TOC3D_EPI_BIAS ? launch_epi<bf16_t, TOC3D_EPI_BIAS>(variant, a, s) : launch_epi<bf16_t, TOC3D_EPI_RESIDUAL>(variant, a, s);
    if (rc || g_bad_variant) { fprintf(stderr, "variant %d refused\n", variant); exit(1); }
}

// one transformer block's four GEMMs (the attention / LayerNorm kernels between them are left out: the question is the GEMM chain)
static void block(const Bufs& b, const Weights& w, int M, const int* v, hipStream_t s) {
    gemm(TOC3D_EPI_BIAS, v[0], b.a, 1024, w.wqkv, 1024, b.qkv, 3072, M, 3072, nullptr, w.bias, s);
    gemm(TOC3D_EPI_RESIDUAL, v[1], b.att, 1024, w.wproj, 1024, b.x, 1024, M, 1024, b.x, w.bias, s);
    gemm(TOC3D_EPI_BIAS, v[2], b.a, 1024, w.w12, 1024, b.hid, 5504, M, 5504, nullptr, w.bias, s);
    gemm(TOC3D_EPI_RESIDUAL, v[3], b.hid, 5504, w.w3, 2752, b.x, 1024, M, 1024, b.x, w.bias, s);
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 24;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("# %s, %d CUs\n", prop.name, ncu);
    const size_t MF = 6000, MH = 3000;
    Bufs full = alloc_bufs(MF + 128), h0 = alloc_bufs(MH + 128), h1 = alloc_bufs(MH + 128);
    Weights w;
    CK(hipMalloc(&w.wqkv, 3072 * 1024 * 2)); CK(hipMalloc(&w.wproj, 1024 * 1024 * 2)); CK(hipMalloc(&w.w12, 5632 * 1024 * 2)); CK(hipMalloc(&w.w3, 1024 * 2752 * 2));
    CK(hipMalloc(&w.bias, 8192 * 4));
    CK(hipMemset(w.wqkv, 0x3c, 3072 * 1024 * 2)); CK(hipMemset(w.wproj, 0x3c, 1024 * 1024 * 2)); CK(hipMemset(w.w12, 0x3c, 5632 * 1024 * 2)); CK(hipMemset(w.w3, 0x3c, 1024 * 2752 * 2));
    CK(hipMemset(w.bias, 0, 8192 * 4));
    // shipped variants at M = 6000 (qkv v52, proj v17, w1|w2 v116, w3 v117); the half-size chains use the same tiles (same tiles per CU on half the CUs)
    const int vf[4] = {52, 17, 116, 117};

    // ---- which CUs does a masked stream get?  bit i of the mask <-> ? ----
    const int words = (ncu + 31) / 32;
    auto make_stream = [&](const std::vector<uint32_t>& mask) { hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data())); return s; };
    std::vector<uint32_t> lowhalf(words, 0), highhalf(words, 0), even(words, 0), odd(words, 0);
    for (int i = 0; i < ncu; ++i) {
        (i < ncu / 2 ? lowhalf : highhalf)[i / 32] |= 1u << (i % 32);
        (i % 2 == 0 ? even : odd)[i / 32] |= 1u << (i % 32);
    }
    unsigned long long* trace;
    CK(hipMalloc(&trace, (1 << 16) * 32));
    unsigned long long* null_ptr = nullptr;
    auto census = [&](const char* name, hipStream_t s) {
        CK(hipMemset(trace, 0, (1 << 16) * 32));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(toc3d_trace_buf), &trace, sizeof(trace)));
        gemm(TOC3D_EPI_BIAS, 16, full.a, 1024, w.wqkv, 1024, full.qkv, 3072, 6000, 3072, nullptr, w.bias, s);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(toc3d_trace_buf), &null_ptr, sizeof(null_ptr)));
        std::vector<unsigned long long> h((1 << 16) * 4);
        CK(hipMemcpy(h.data(), trace, (1 << 16) * 32, hipMemcpyDeviceToHost));
        std::map<int, std::set<int>> per_xcc;
        for (size_t i = 0; i < (1 << 16); ++i) {
            if (!h[4 * i]) continue;
            const unsigned long long id = h[4 * i + 3];
            const unsigned hw = (unsigned)id;
            per_xcc[(int)(id >> 32)].insert((int)(((hw >> 13) & 7) << 8 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 0xf)));
        }
        printf("%-26s CUs used per XCC:", name);
        int tot = 0;
        for (auto& kv : per_xcc) { printf(" x%d:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
        printf("  (total %d)\n", tot);
    };
    hipStream_t s_plain, s_lo = make_stream(lowhalf), s_hi = make_stream(highhalf), s_ev = make_stream(even), s_od = make_stream(odd);
    CK(hipStreamCreateWithFlags(&s_plain, hipStreamNonBlocking));
    hipStream_t s_plain2;
    CK(hipStreamCreateWithFlags(&s_plain2, hipStreamNonBlocking));
    census("unmasked", s_plain);
    census("mask bits [0, n/2)", s_lo);
    census("mask bits [n/2, n)", s_hi);
    census("mask even bits", s_ev);
    census("mask odd bits", s_od);

    auto wall = [&](auto&& fn, int reps) {
        fn();
        CK(hipDeviceSynchronize());
        double best = 1e30;
        for (int r = 0; r < reps; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            fn();
            CK(hipDeviceSynchronize());
            best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        return best;
    };
    const double t_full = wall([&] { for (int i = 0; i < blocks; ++i) block(full, w, (int)MF, vf, s_plain); }, 7);
    const double t_half1 = wall([&] { for (int i = 0; i < blocks; ++i) block(h0, w, (int)MH, vf, s_plain); }, 7);
    const double t_two = wall([&] { for (int i = 0; i < blocks; ++i) { block(h0, w, (int)MH, vf, s_plain); block(h1, w, (int)MH, vf, s_plain2); } }, 7);
    const double t_lohi = wall([&] { for (int i = 0; i < blocks; ++i) { block(h0, w, (int)MH, vf, s_lo); block(h1, w, (int)MH, vf, s_hi); } }, 7);
    const double t_evod = wall([&] { for (int i = 0; i < blocks; ++i) { block(h0, w, (int)MH, vf, s_ev); block(h1, w, (int)MH, vf, s_od); } }, 7);
    const double t_lo_alone = wall([&] { for (int i = 0; i < blocks; ++i) block(h0, w, (int)MH, vf, s_lo); }, 7);
    // the same with the second chain offset by half a block (out of phase from the start)
    const double t_lohi_skew = wall([&] {
        gemm(TOC3D_EPI_BIAS, vf[0], h1.a, 1024, w.wqkv, 1024, h1.qkv, 3072, (int)MH, 3072, nullptr, w.bias, s_hi);
        gemm(TOC3D_EPI_RESIDUAL, vf[1], h1.att, 1024, w.wproj, 1024, h1.x, 1024, (int)MH, 1024, h1.x, w.bias, s_hi);
        for (int i = 0; i < blocks; ++i) { block(h0, w, (int)MH, vf, s_lo); block(h1, w, (int)MH, vf, s_hi); } }, 7);
    const double gf = blocks * 2.0 * 6000 * 1024.0 * (3072 + 1024 + 5504 + 2752) * 1e-6;   // MFLOP of the full chain
    printf("%d blocks x 4 GEMMs (qkv, proj, w1|w2, w3), bf16, host wall time of the whole chain (best of 7):\n", blocks);
    printf("  one stream, M = 6000                                  %8.1f us   %5.0f TF\n", t_full, gf / t_full);
    printf("  one stream, M = 3000 (half the work)                  %8.1f us   %5.0f TF\n", t_half1, gf / 2 / t_half1);
    printf("  two unmasked streams, M = 3000 each                   %8.1f us   %5.0f TF\n", t_two, gf / t_two);
    printf("  two streams on complementary CU halves (low | high)   %8.1f us   %5.0f TF\n", t_lohi, gf / t_lohi);
    printf("  ... second chain started half a block early           %8.1f us   %5.0f TF\n", t_lohi_skew, gf / t_lohi_skew);
    printf("  two streams on even | odd mask bits                   %8.1f us   %5.0f TF\n", t_evod, gf / t_evod);
    printf("  one masked stream alone (low half), M = 3000          %8.1f us   %5.0f TF\n", t_lo_alone, gf / 2 / t_lo_alone);
    return 0;
}
