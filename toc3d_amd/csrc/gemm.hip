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
#include "capi.h"
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128;          // ROWB: bytes of K per LDS row per stage
constexpr int TILE_BYTES = BM * ROWB;                   // 16 KiB per operand per stage
constexpr int GEMM_LDS = 2 * 2 * TILE_BYTES;            // 2 stages x (A + B) = 64 KiB

struct GemmArgs {
    const void* A; int64_t lda;
    const void* W; int64_t ldw;
    const float* bias;
    void* out; int64_t ldo;
    const float* res; int64_t ldr; int res_mod;
    float* rep_out; int rep_period;
    int M, N, K, n_valid;
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// stage one 128-row x 128-byte operand tile: 1024 16-byte chunks, 4 per thread.
template <typename T>
TOC3D_DEV void stage_tile(const T* __restrict__ g, int64_t ld, int row0, int max_row, int k0, char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int cidx = t * 256 + wave * 64 + lane;
        const int r = cidx >> 3, p = cidx & 7;
        int gr = row0 + r;
        gr = gr < max_row ? gr : max_row;
        const char* src = reinterpret_cast<const char*>(g + (int64_t)gr * ld + k0) + ((p ^ (r & 7)) << 4);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_tile + (t * 256 + wave * 64) * 16), 16, 0, 0);
    }
}

template <typename T> struct KSteps;                    // 32-wide K steps per 128-byte stage row
template <> struct KSteps<bf16_t> { static constexpr int n = 2; };
template <> struct KSteps<float> { static constexpr int n = 1; };

// fragment of row r (tile-local) for K step s, lane group g = lane >> 4
TOC3D_DEV Frag<bf16_t> lds_frag(const char* tile, int r, int s, int g, bf16_t) {
    const int cc = s * 4 + g;
    Frag<bf16_t> f;
    f.v = *reinterpret_cast<const bf16x8*>(tile + r * ROWB + ((cc ^ (r & 7)) << 4));
    return f;
}
TOC3D_DEV Frag<float> lds_frag(const char* tile, int r, int /*s*/, int g, float) {
    Frag<float> f;
    f.lo = *reinterpret_cast<const f32x4*>(tile + r * ROWB + (((2 * g) ^ (r & 7)) << 4));
    f.hi = *reinterpret_cast<const f32x4*>(tile + r * ROWB + (((2 * g + 1) ^ (r & 7)) << 4));
    return f;
}

TOC3D_DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
TOC3D_DEV float silu(float x) { return x / (1.0f + expf(-x)); }

template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, g = lane >> 4;

    const int tiles_n = (a.N + BN - 1) / BN;
    const int tiles_m = (a.M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

    const T* A = reinterpret_cast<const T*>(a.A);
    const T* W = reinterpret_cast<const T*>(a.W);
    constexpr int BK = ROWB / (int)sizeof(T);
    const int nk = a.K / BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // W rows are padded to a multiple of 128 at pack time, A rows are clamped to M-1
    const int w_max = tiles_n * BN - 1;
    stage_tile<T>(A, a.lda, m0, a.M - 1, 0, smem, wave, lane);
    stage_tile<T>(W, a.ldw, n0, w_max, 0, smem + TILE_BYTES, wave, lane);
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        char* sA = smem + cur * 2 * TILE_BYTES;
        char* sB = sA + TILE_BYTES;
        if (kt + 1 < nk) {
            char* nA = smem + (cur ^ 1) * 2 * TILE_BYTES;
            stage_tile<T>(A, a.lda, m0, a.M - 1, (kt + 1) * BK, nA, wave, lane);
            stage_tile<T>(W, a.ldw, n0, w_max, (kt + 1) * BK, nA + TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int s = 0; s < KSteps<T>::n; ++s) {
            Frag<T> fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = lds_frag(sA, wm * 64 + i * 16 + r16, s, g, T());
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = lds_frag(sB, wn * 64 + j * 16 + r16, s, g, T());
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma_step(acc[i][j], fa[i], fb[j]);
        }
        __syncthreads();          // drains the in-flight global_load_lds (vmcnt(0)) and frees `cur`
        cur ^= 1;
    }

    // ---- epilogue: lane holds C[row = .. + g*4 + r][col = .. + r16] ----
    if (EPI == TOC3D_EPI_SWIGLU) {
        // packed columns: per 32-column group, cols 0-15 = w1 units, cols 16-31 = w2 of the same units
        T* out = reinterpret_cast<T*>(a.out);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const int pc = n0 + wn * 64 + jp * 32 + r16;      // packed col of the w1 half
                const int unit = (pc >> 5) * 16 + r16;
                if (pc < a.N) {
                    const float b1 = a.bias[pc], b2 = a.bias[pc + 16];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + wm * 64 + i * 16 + g * 4 + r;
                        if (row < a.M) {
                            const float x1 = acc[i][2 * jp][r] + b1, x2 = acc[i][2 * jp + 1][r] + b2;
                            const float h = unit < a.n_valid ? silu(x1) * x2 : 0.f;
                            out[(int64_t)row * a.ldo + unit] = to_act<T>(h);
                        }
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + r16;
            if (col >= a.N) continue;
            const float b = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + g * 4 + r;
                if (row >= a.M) continue;
                const float raw = acc[i][j][r] + b;
                if (EPI == TOC3D_EPI_BIAS) {
                    reinterpret_cast<T*>(a.out)[(int64_t)row * a.ldo + col] = to_act<T>(raw);
                } else if (EPI == TOC3D_EPI_GELU) {
                    reinterpret_cast<T*>(a.out)[(int64_t)row * a.ldo + col] = to_act<T>(gelu_erf(raw));
                } else {   // TOC3D_EPI_RESIDUAL: f32 out = residual + (acc + bias)
                    const int rr = a.res_mod > 0 ? row % a.res_mod : row;
                    const float base = a.res ? a.res[(int64_t)rr * a.ldr + col] : 0.f;
                    reinterpret_cast<float*>(a.out)[(int64_t)row * a.ldo + col] = base + raw;
                    if (a.rep_period > 0 && (row % a.rep_period) == a.rep_period - 1)
                        a.rep_out[(int64_t)(row / a.rep_period) * a.N + col] = raw;
                }
            }
        }
    }
}

template <typename T, int EPI>
void launch_one(dim3 grid, const GemmArgs& a, hipStream_t s) {
    static bool attr_set = false;      // 64 KiB of dynamic LDS: raise the per-kernel limit once
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<T, EPI>), grid, dim3(256), GEMM_LDS, s, a);
}

template <typename T>
int launch_gemm(int epi, const GemmArgs& a, hipStream_t s) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    dim3 grid(tiles);
    switch (epi) {
        case TOC3D_EPI_BIAS: launch_one<T, TOC3D_EPI_BIAS>(grid, a, s); break;
        case TOC3D_EPI_RESIDUAL: launch_one<T, TOC3D_EPI_RESIDUAL>(grid, a, s); break;
        case TOC3D_EPI_SWIGLU: launch_one<T, TOC3D_EPI_SWIGLU>(grid, a, s); break;
        case TOC3D_EPI_GELU: launch_one<T, TOC3D_EPI_GELU>(grid, a, s); break;
        default: return TOC3D_ERR_ARG;
    }
    return TOC3D_OK;
}

// ---- weight packing ---------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, int N, int K, T* __restrict__ out, int Np, int Kp) {
    const int64_t total = (int64_t)Np * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / Kp), k = (int)(i % Kp);
        out[i] = to_act<T>((n < N && k < K) ? w[(int64_t)n * K + k] : 0.f);
    }
}

// interleave mlp.w1 / mlp.w2 so the SwiGLU epilogue finds x1 and x2 of one hidden unit in one lane:
// packed row 32*b + i      = w1 row 16*b + i   (i < 16)
// packed row 32*b + 16 + i = w2 row 16*b + i
template <typename T>
__global__ void pack_swiglu_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ b1,
                                   const float* __restrict__ b2, int Hd, int K, T* __restrict__ out_w, float* __restrict__ out_b,
                                   int Hp, int Kp) {
    const int64_t total = (int64_t)2 * Hp * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int pr = (int)(i / Kp), k = (int)(i % Kp);
        const int unit = (pr >> 5) * 16 + (pr & 15);
        const bool second = (pr & 16) != 0;
        float v = 0.f;
        if (unit < Hd && k < K) v = (second ? w2 : w1)[(int64_t)unit * K + k];
        out_w[i] = to_act<T>(v);
        if (k == 0) out_b[pr] = unit < Hd ? (second ? b2 : b1)[unit] : 0.f;
    }
}

// im2col for the k = s = patch conv: row m = (v, pr, pc), col kk = (ch, py, px) -- matches the
// flattened Conv2d weight (C, Cin, p, p), eva_utils.py:279-281.
template <typename T>
__global__ void im2col_kernel(const float* __restrict__ img, T* __restrict__ out, int64_t ldo, int V, int Cin, int H, int W, int p) {
    const int h = H / p, w = W / p;
    const int Kc = Cin * p * p;
    const int64_t total = (int64_t)V * h * w * Kc / 4;       // 4 consecutive px per thread (p % 4 == 0)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i % (Kc / 4)) * 4;
        const int64_t m = i / (Kc / 4);
        const int px = kk % p, py = (kk / p) % p, ch = kk / (p * p);
        const int pc = (int)(m % w), pr = (int)((m / w) % h), v = (int)(m / ((int64_t)w * h));
        const float* src = img + (((int64_t)v * Cin + ch) * H + pr * p + py) * W + pc * p + px;
        const f32x4 x = *reinterpret_cast<const f32x4*>(src);
        T* dst = out + m * ldo + kk;
        dst[0] = to_act<T>(x[0]); dst[1] = to_act<T>(x[1]); dst[2] = to_act<T>(x[2]); dst[3] = to_act<T>(x[3]);
    }
}

}  // namespace

extern "C" {

int toc3d_linear(int dtype, int epilogue, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                 void* out, int64_t ldo, const float* residual, int64_t ldr, int64_t residual_row_mod,
                 float* rep_out, int64_t rep_period, int64_t M, int64_t N, int64_t K, int64_t n_valid,
                 toc3d_stream_t stream) {
    TOC3D_REQUIRE(dtype == TOC3D_F32 || dtype == TOC3D_BF16, "toc3d_linear: bad dtype %d", dtype);
    TOC3D_REQUIRE(A && W && out, "toc3d_linear: null buffer");
    TOC3D_REQUIRE(M >= 0 && N > 0 && K > 0, "toc3d_linear: bad dims M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    const int bk = dtype == TOC3D_BF16 ? 64 : 32;
    TOC3D_REQUIRE(K % bk == 0, "toc3d_linear: K=%lld must be a multiple of %d (pad at pack time)", (long long)K, bk);
    TOC3D_REQUIRE(lda >= K && ldw >= K, "toc3d_linear: leading dims smaller than K");
    TOC3D_REQUIRE((lda * (dtype == TOC3D_BF16 ? 2 : 4)) % 16 == 0 && (ldw * (dtype == TOC3D_BF16 ? 2 : 4)) % 16 == 0,
                  "toc3d_linear: rows must be 16-byte aligned");
    TOC3D_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "toc3d_linear: A/W must be 16-byte aligned");
    if (epilogue == TOC3D_EPI_SWIGLU) {
        TOC3D_REQUIRE(bias && N % 32 == 0 && n_valid > 0 && n_valid <= N / 2, "toc3d_linear: swiglu needs bias, N%%32==0, n_valid");
        TOC3D_REQUIRE(ldo >= N / 2, "toc3d_linear: swiglu ldo < N/2");
    } else {
        TOC3D_REQUIRE(ldo >= N, "toc3d_linear: ldo < N");
    }
    if (epilogue == TOC3D_EPI_RESIDUAL) {
        TOC3D_REQUIRE(!residual || ldr >= N, "toc3d_linear: ldr < N");
        TOC3D_REQUIRE(rep_period == 0 || rep_out, "toc3d_linear: rep_period set without rep_out");
    }
    if (M == 0) return TOC3D_OK;
    GemmArgs a{A, lda, W, ldw, bias, out, ldo, residual, ldr, (int)residual_row_mod, rep_out, (int)rep_period,
               (int)M, (int)N, (int)K, (int)n_valid};
    int rc = dtype == TOC3D_BF16 ? launch_gemm<bf16_t>(epilogue, a, as_stream(stream)) : launch_gemm<float>(epilogue, a, as_stream(stream));
    if (rc != TOC3D_OK) { toc3d_set_error("toc3d_linear: bad epilogue %d", epilogue); return rc; }
    TOC3D_LAUNCH_CHECK("toc3d_linear");
    return TOC3D_OK;
}

int toc3d_pack_weight(int dtype, const float* w, int64_t N, int64_t K, void* out, int64_t Np, int64_t Kp, toc3d_stream_t stream) {
    TOC3D_REQUIRE(w && out && Np >= N && Kp >= K && N > 0 && K > 0, "toc3d_pack_weight: bad arguments");
    const int blocks = (int)((Np * Kp + 255) / 256 < 4096 ? (Np * Kp + 255) / 256 : 4096);
    if (dtype == TOC3D_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), w, (int)N, (int)K, (bf16_t*)out, (int)Np, (int)Kp);
    else if (dtype == TOC3D_F32)
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), w, (int)N, (int)K, (float*)out, (int)Np, (int)Kp);
    else { toc3d_set_error("toc3d_pack_weight: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_pack_weight");
    return TOC3D_OK;
}

int toc3d_pack_swiglu(int dtype, const float* w1, const float* w2, const float* b1, const float* b2, int64_t Hd, int64_t K,
                      void* out_w, float* out_b, int64_t Hp, int64_t Kp, toc3d_stream_t stream) {
    TOC3D_REQUIRE(w1 && w2 && b1 && b2 && out_w && out_b, "toc3d_pack_swiglu: null buffer");
    TOC3D_REQUIRE(Hp >= Hd && Hp % 64 == 0 && Kp >= K, "toc3d_pack_swiglu: Hp must be >= Hd and a multiple of 64");
    const int64_t total = 2 * Hp * Kp;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == TOC3D_BF16)
        hipLaunchKernelGGL(pack_swiglu_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), w1, w2, b1, b2, (int)Hd, (int)K, (bf16_t*)out_w, out_b, (int)Hp, (int)Kp);
    else if (dtype == TOC3D_F32)
        hipLaunchKernelGGL(pack_swiglu_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), w1, w2, b1, b2, (int)Hd, (int)K, (float*)out_w, out_b, (int)Hp, (int)Kp);
    else { toc3d_set_error("toc3d_pack_swiglu: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_pack_swiglu");
    return TOC3D_OK;
}

int toc3d_im2col_patches(int dtype, const float* img, void* out, int64_t ldo, int64_t V, int64_t Cin, int64_t H, int64_t W,
                         int64_t patch, toc3d_stream_t stream) {
    TOC3D_REQUIRE(img && out, "toc3d_im2col_patches: null buffer");
    TOC3D_REQUIRE(patch > 0 && patch % 4 == 0 && H % patch == 0 && W % patch == 0, "toc3d_im2col_patches: H, W must be multiples of patch (patch %% 4 == 0)");
    TOC3D_REQUIRE(ldo >= Cin * patch * patch, "toc3d_im2col_patches: ldo too small");
    TOC3D_REQUIRE(((uintptr_t)img % 16) == 0, "toc3d_im2col_patches: img must be 16-byte aligned");
    const int64_t total = V * (H / patch) * (W / patch) * Cin * patch * patch / 4;
    if (total == 0) return TOC3D_OK;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == TOC3D_BF16)
        hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), img, (bf16_t*)out, ldo, (int)V, (int)Cin, (int)H, (int)W, (int)patch);
    else if (dtype == TOC3D_F32)
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), img, (float*)out, ldo, (int)V, (int)Cin, (int)H, (int)W, (int)patch);
    else { toc3d_set_error("toc3d_im2col_patches: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_im2col_patches");
    return TOC3D_OK;
}

}  // extern "C"
