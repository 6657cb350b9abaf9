// Development tool (not part of the library): per-workgroup timeline of one GEMM launch on MI355X.
//
// Builds its own copy of the kernels of toc3d_amd/csrc/gemm_kernels.h with -DTOC3D_GEMM_TRACE: every workgroup leaves the 100 MHz
// real-time counter at entry, after its K loop and after its stores were acknowledged, plus the id of the CU it ran on.  The report
// answers what a launch's time is made of: dispatch ramp, the K loops, the epilogues, the tail behind the last full round.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTOC3D_GEMM_TRACE -Itoc3d_amd/csrc tools/ubench/gemm_timeline.hip -o gpurun_out/gemm_timeline
//   gpurun_out/gemm_timeline [cold=1]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <vector>

#include "gemm_kernels.h"

thread_local bool g_bad_variant = false;
thread_local Toc3dPlan* toc3d_tls_recording = nullptr;
void toc3d_plan_record(Toc3dPlan*, const void*, dim3, dim3, size_t, hipStream_t, const void* const*, const size_t*, const size_t*, int) {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape { const char* name; int epi, M, N, K, variant; };

static int launch(const Shape& sh, const GemmArgs& a, hipStream_t s) {
    g_bad_variant = false;
    int rc;
    if (sh.epi == TOC3D_EPI_BIAS) rc = launch_epi<bf16_t, TOC3D_EPI_BIAS>(sh.variant, a, s);
    else if (sh.epi == TOC3D_EPI_QKV_ROPE) rc = launch_epi<bf16_t, TOC3D_EPI_QKV_ROPE>(sh.variant, a, s);
    else rc = launch_epi<bf16_t, TOC3D_EPI_RESIDUAL>(sh.variant, a, s);
    return rc != 0 || g_bad_variant;
}

int main(int argc, char** argv) {
    const bool cold = argc > 1 && atoi(argv[1]) != 0;
    const bool rope_only = argc > 2 && atoi(argv[2]) != 0;
    // the frame's GEMM classes with the shipped tile variants (toc3d_amd/tuned/toc3d_faster_320x800_bf16.json); w1|w2 with the plain
    // bias epilogue and w3 with the plain residual epilogue (the folded-LayerNorm epilogues add their own phases, not traced here)
    const Shape rope_shapes[] = {
        {"qkv  M=6000 bias", TOC3D_EPI_BIAS, 6000, 3072, 1024, 52},  {"qkv  M=6000 rope", TOC3D_EPI_QKV_ROPE, 6000, 3072, 1024, 52},
        {"qkv  M=3744 bias", TOC3D_EPI_BIAS, 3744, 3072, 1024, 49},  {"qkv  M=3744 rope", TOC3D_EPI_QKV_ROPE, 3744, 3072, 1024, 49},
        {"qkv  M=3276 bias", TOC3D_EPI_BIAS, 3276, 3072, 1024, 45},  {"qkv  M=3276 rope", TOC3D_EPI_QKV_ROPE, 3276, 3072, 1024, 45},
        {"qkv  M=2808 bias", TOC3D_EPI_BIAS, 2808, 3072, 1024, 53},  {"qkv  M=2808 rope", TOC3D_EPI_QKV_ROPE, 2808, 3072, 1024, 53},
        {"qkv  M=2178 bias", TOC3D_EPI_BIAS, 2178, 3072, 1024, 17},  {"qkv  M=2178 rope", TOC3D_EPI_QKV_ROPE, 2178, 3072, 1024, 17},
        {"qkv  M=3276 v16 bias", TOC3D_EPI_BIAS, 3276, 3072, 1024, 16},  {"qkv  M=3276 v16 rope", TOC3D_EPI_QKV_ROPE, 3276, 3072, 1024, 16},
    };
    const Shape all_shapes[] = {
        {"qkv  M=6000", TOC3D_EPI_BIAS, 6000, 3072, 1024, 52},      {"qkv  M=3276", TOC3D_EPI_BIAS, 3276, 3072, 1024, 45},
        {"qkv  M=2808", TOC3D_EPI_BIAS, 2808, 3072, 1024, 53},      {"w12  M=6000", TOC3D_EPI_BIAS, 6000, 5504, 1024, 116},
        {"w12  M=3276", TOC3D_EPI_BIAS, 3276, 5504, 1024, 16},      {"w12  M=2808", TOC3D_EPI_BIAS, 2808, 5504, 1024, 19},
        {"w12  M=2808 v60", TOC3D_EPI_BIAS, 2808, 5504, 1024, 60},  {"qkv  M=6000 v60", TOC3D_EPI_BIAS, 6000, 3072, 1024, 60},
        {"proj M=6000", TOC3D_EPI_RESIDUAL, 6000, 1024, 1024, 17},  {"proj M=3276", TOC3D_EPI_RESIDUAL, 3276, 1024, 1024, 26},
        {"proj M=2808", TOC3D_EPI_RESIDUAL, 2808, 1024, 1024, 114}, {"w3   M=6000", TOC3D_EPI_RESIDUAL, 6000, 1024, 2752, 117},
        {"w3   M=3276", TOC3D_EPI_RESIDUAL, 3276, 1024, 2752, 14},  {"w3   M=2808", TOC3D_EPI_RESIDUAL, 2808, 1024, 2752, 114},
    };
    const size_t MAXM = 6016, MAXN = 5632, MAXK = 2752;
    bf16_t *A, *W, *outb;
    float *bias, *outf;
    CK(hipMalloc(&A, MAXM * MAXK * 2)); CK(hipMalloc(&W, MAXN * MAXK * 2)); CK(hipMalloc(&outb, MAXM * MAXN * 2));
    CK(hipMalloc(&outf, MAXM * 1024 * 4)); CK(hipMalloc(&bias, MAXN * 4));
    {   // uniform random [-1, 1) operands (zero-filled operands clock ~15 % higher: cdna_hip_programming.md rule 25)
        std::vector<uint16_t> h(MAXM * MAXK);
        uint32_t x = 12345u;
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; const float f = (float)(x >> 8) / 8388608.0f - 1.0f; uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
        for (auto& v : h) v = rnd();
        CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        h.resize(MAXN * MAXK);
        for (auto& v : h) v = rnd();
        CK(hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(bias, 0, MAXN * 4)); CK(hipMemset(outf, 0, MAXM * 1024 * 4));
    }
    char* flush = nullptr;
    const size_t FLUSH = 512u << 20;
    if (cold) CK(hipMalloc(&flush, FLUSH));
    unsigned long long* trace;
    const size_t TRACE_WG = 1 << 16;
    CK(hipMalloc(&trace, TRACE_WG * 32));
    unsigned long long* null_ptr = nullptr;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# per-workgroup timeline (100 MHz real-time counter), %s operands/caches; times in us\n", cold ? "COLD (512 MB memset in front of every launch)" : "warm");
    printf("# name | variant | WGs | event us | span = first entry -> last exit | ramp = last entry of the first wave of WGs (all that start within 1 us x CUs) | WG: K-loop median / epilogue median / total median, max | tail = span - time when half the CUs went idle\n");
    int32_t* rc_d; float* tab_d;
    CK(hipMalloc(&rc_d, MAXM * 4)); CK(hipMalloc(&tab_d, 2 * 2 * 20 * 16 * 4));
    {
        std::vector<int32_t> rc(MAXM);
        for (size_t i = 0; i < MAXM; ++i) rc[i] = (int)(((i / 50) % 20) << 16 | ((i % 50) % 20));
        CK(hipMemcpy(rc_d, rc.data(), MAXM * 4, hipMemcpyHostToDevice));
        std::vector<float> tab(2 * 2 * 20 * 16);
        for (size_t i = 0; i < tab.size(); ++i) tab[i] = (i < tab.size() / 2) ? cosf(0.01f * i) : sinf(0.01f * i);
        CK(hipMemcpy(tab_d, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    }
    const Shape* shapes = rope_only ? rope_shapes : all_shapes;
    const size_t nshapes = rope_only ? sizeof(rope_shapes) / sizeof(Shape) : sizeof(all_shapes) / sizeof(Shape);
    for (size_t si = 0; si < nshapes; ++si) {
        const Shape& sh = shapes[si];
        GemmArgs a{};
        a.rope_rc = rc_d; a.rope_tab = tab_d; a.rope_L = 20; a.rope_scale = 0.125f;
        a.A = A; a.lda = sh.K; a.W = W; a.ldw = sh.K; a.bias = bias;
        const bool res = sh.epi == TOC3D_EPI_RESIDUAL;
        a.out = res ? (void*)outf : (void*)outb; a.ldo = sh.N;
        a.res = res ? outf : nullptr; a.ldr = sh.N;
        a.M = sh.M; a.N = sh.N; a.K = sh.K; a.vec = 1;
        for (int i = 0; i < 3; ++i)
            if (launch(sh, a, s)) { printf("%s: variant %d refused\n", sh.name, sh.variant); break; }
        CK(hipStreamSynchronize(s));
        // un-traced timing
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            if (cold) CK(hipMemsetAsync(flush, rep, FLUSH, s));
            CK(hipEventRecord(e0, s));
            launch(sh, a, s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms * 1000.f);
        }
        // traced launch
        CK(hipMemset(trace, 0, TRACE_WG * 32));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(toc3d_trace_buf), &trace, sizeof(trace)));
        if (cold) CK(hipMemsetAsync(flush, 7, FLUSH, s));
        launch(sh, a, s);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(toc3d_trace_buf), &null_ptr, sizeof(null_ptr)));
        std::vector<unsigned long long> h(TRACE_WG * 4);
        CK(hipMemcpy(h.data(), trace, TRACE_WG * 32, hipMemcpyDeviceToHost));
        std::vector<double> t0, t1, t2;
        std::map<unsigned long long, int> cus;
        unsigned long long first = ~0ull, last = 0;
        for (size_t i = 0; i < TRACE_WG; ++i) {
            if (!h[4 * i] || !h[4 * i + 2]) continue;
            first = std::min(first, h[4 * i]); last = std::max(last, h[4 * i + 2]);
        }
        for (size_t i = 0; i < TRACE_WG; ++i) {
            if (!h[4 * i] || !h[4 * i + 2]) continue;
            t0.push_back((h[4 * i] - first) * 0.01); t1.push_back((h[4 * i + 1] - first) * 0.01); t2.push_back((h[4 * i + 2] - first) * 0.01);
            const unsigned long long id = h[4 * i + 3];
            const unsigned hw = (unsigned)id;
            cus[((id >> 32) << 16) | ((hw >> 8) & 0xff) | (((hw >> 13) & 7) << 8)]++;      // xcc | se | sh, cu
        }
        const size_t n = t0.size();
        if (!n) { printf("%s: no trace\n", sh.name); continue; }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        std::vector<double> kl(n), ep(n), tot(n);
        for (size_t i = 0; i < n; ++i) { kl[i] = t1[i] - t0[i]; ep[i] = t2[i] - t1[i]; tot[i] = t2[i] - t0[i]; }
        // concurrency over time: sweep; time at which the number of running WGs last drops below half its peak
        std::vector<std::pair<double, int>> evs;
        for (size_t i = 0; i < n; ++i) { evs.push_back({t0[i], 1}); evs.push_back({t2[i], -1}); }
        std::sort(evs.begin(), evs.end());
        int cur = 0, peak = 0;
        for (auto& e : evs) { cur += e.second; peak = std::max(peak, cur); }
        cur = 0;
        double t_half = 0;
        for (auto& e : evs) { const int before = cur; cur += e.second; if (before >= peak / 2 && cur < peak / 2) t_half = e.first; }
        std::vector<double> starts = t0;
        std::sort(starts.begin(), starts.end());
        const double span = (last - first) * 0.01;
        const double flops = 2.0 * sh.M * sh.N * sh.K;
        printf("%-16s | v%-3d | %5zu WGs on %3zu CUs, peak %4d concurrent | event %6.1f us (%4.0f TF) | span %6.1f | start p50 %5.1f p90 %5.1f max %5.1f | K-loop med %5.1f | epilogue med %5.1f | WG total med %5.1f max %5.1f | half-idle at %6.1f (tail %5.1f)\n",
               sh.name, sh.variant, n, cus.size(), peak, best, flops / best * 1e-6, span, starts[n / 2], starts[n * 9 / 10], starts[n - 1], med(kl), med(ep), med(tot),
               *std::max_element(tot.begin(), tot.end()), t_half, span - t_half);
        // coarse histogram of running workgroups over the span (10 bins)
        printf("    running WGs by tenth of the span:");
        for (int b = 0; b < 10; ++b) {
            const double t = span * (b + 0.5) / 10;
            int c = 0;
            for (size_t i = 0; i < n; ++i) c += t0[i] <= t && t < t2[i];
            printf(" %4d", c);
        }
        printf("\n");
    }
    return 0;
}
