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
// This file: the C ABI of the linear layers, weight packing and the im2col / image kernels.  The GEMM kernels live in gemm_kernels.h and are
// instantiated by gemm_epi_{plain,residual,swiglu}.hip.
#include <cstdlib>

#include "gemm_kernels.h"

thread_local bool g_bad_variant = false;               // variant cannot serve the requested epilogue

namespace {

int launch_gemm(int is_bf16, int epi, int variant, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case TOC3D_EPI_BIAS: case TOC3D_EPI_GELU: case TOC3D_EPI_CONV3X3: return toc3d_gemm_launch_plain(is_bf16, epi, variant, a, s);
        case TOC3D_EPI_RESIDUAL: case TOC3D_EPI_RESIDUAL_LN: case TOC3D_EPI_RESIDUAL_STATS: return toc3d_gemm_launch_residual(is_bf16, epi, variant, a, s);
        case TOC3D_EPI_SWIGLU: case TOC3D_EPI_SWIGLU_STATS: case TOC3D_EPI_SWIGLU_STATS_LN: return toc3d_gemm_launch_swiglu(is_bf16, epi, variant, a, s);
        case TOC3D_EPI_QKV_ROPE: return toc3d_gemm_launch_rope(is_bf16, epi, variant, a, s);
        default: return TOC3D_ERR_ARG;
    }
}

// ---- weight packing ---------------------------------------------------------------------------------
// f32 rows -> (hi, lo) bf16 planes (gemm_kernels.h, store_planes4): eight lanes per 32-element group, 16 bytes in, 8 + 8 bytes out per lane.  In place is fine:
// a group is read by one load instruction of one wavefront before that wavefront's stores can issue.
__global__ __launch_bounds__(256) void x3_planes_kernel(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int gpr) {
    const int64_t grp = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (grp >= rows * gpr) return;
    const int64_t row = grp / gpr;
    const int g = (int)(grp % gpr), q = threadIdx.x & 7;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + row * ld_src + g * 32 + q * 4);
    const float x[4] = {v[0], v[1], v[2], v[3]};
    store_planes4(dst + row * ld_dst, g * 32 + q * 4, x);
}

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

// gamma-scaled w3 + the two column vectors of the folded ffn_ln (include/toc3d.h, toc3d_pack_weight_lnfold): one workgroup per output
// row n; c1 sums the ROUNDED scaled weights, c2 = beta . w3[n] + b3[n]; fixed-order tree reduction.
template <typename T>
__global__ __launch_bounds__(256) void pack_lnfold_kernel(const float* __restrict__ w3, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ b3, int N, int K, T* __restrict__ out, int Kp,
                                                          float* __restrict__ c1, float* __restrict__ c2) {
    __shared__ float r1[256], r2[256];
    const int n = blockIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int k = threadIdx.x; k < Kp; k += 256) {
        T v = to_act<T>(0.f);
        if (n < N && k < K) {
            const float w = w3[(int64_t)n * K + k];
            v = to_act<T>(gamma[k] * w);
            s1 += from_act(v);
            s2 = __builtin_fmaf(beta[k], w, s2);
        }
        out[(int64_t)n * Kp + k] = v;
    }
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0 && n < N) { c1[n] = r1[0]; c2[n] = r2[0] + b3[n]; }
}

// norm2 folded into the w1|w2 GEMM: the interleaved SwiGLU weights scaled by gamma per input channel, c1 = row sums of the ROUNDED scaled
// weights, c2 = beta . w + b, both in packed row order (toc3d_pack_swiglu_lnfold).  One workgroup per packed row.
template <typename T>
__global__ __launch_bounds__(256) void pack_swiglu_lnfold_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ b1,
                                                                 const float* __restrict__ b2, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 int Hd, int K, T* __restrict__ out_w, float* __restrict__ c1, float* __restrict__ c2, int Kp) {
    __shared__ float r1[256], r2[256];
    const int pr = blockIdx.x;
    const int unit = (pr >> 5) * 16 + (pr & 15);
    const bool second = (pr & 16) != 0;
    const float* w = (second ? w2 : w1) + (int64_t)unit * K;
    float s1 = 0.f, s2 = 0.f;
    for (int k = threadIdx.x; k < Kp; k += 256) {
        T v = to_act<T>(0.f);
        if (unit < Hd && k < K) {
            v = to_act<T>(gamma[k] * w[k]);
            s1 += from_act(v);
            s2 = __builtin_fmaf(beta[k], w[k], s2);
        }
        out_w[(int64_t)pr * Kp + k] = v;
    }
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { c1[pr] = r1[0]; c2[pr] = unit < Hd ? r2[0] + (second ? b2 : b1)[unit] : 0.f; }
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

// ---- uint8 camera images: NormalizeMultiviewImage + PadMultiViewImage + HWC->CHW, standalone or fused
// into the patch-embedding im2col.  Arithmetic exactly as mmcv.imnormalize on a CV_32F image (oracle/image_oracle.py):
// fl32(fl32(x - mean) * stdinv), channel flip first when to_rgb, padded pixels exactly 0.
struct ImgNorm { float mean[3]; float stdinv[3]; int to_rgb; };

TOC3D_DEV float norm_px(unsigned v, int c, const ImgNorm& n) {
    const float d = __fsub_rn((float)v, n.mean[c]);
    return __fmul_rn(d, n.stdinv[c]);
}

__global__ void normalize_images_kernel(const uint8_t* __restrict__ img, int V, int H, int W, ImgNorm n, float* __restrict__ out, int Hp, int Wp) {
    const int64_t total = (int64_t)V * Hp * (Wp / 4);                    // 4 consecutive px (all 3 channels) per thread
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x0 = (int)(i % (Wp / 4)) * 4, y = (int)((i / (Wp / 4)) % Hp), v = (int)(i / ((int64_t)(Wp / 4) * Hp));
        float o[3][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = y < H && x0 + q < W;
            const uint8_t* px = img + (((int64_t)v * H + y) * W + x0 + q) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c][q] = in ? norm_px(px[n.to_rgb ? 2 - c : c], c, n) : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            *reinterpret_cast<f32x4*>(out + (((int64_t)v * 3 + c) * Hp + y) * Wp + x0) = f32x4{o[c][0], o[c][1], o[c][2], o[c][3]};
    }
}

template <typename T>
__global__ void im2col_u8_kernel(const uint8_t* __restrict__ img, int V, int H, int W, ImgNorm n, T* __restrict__ out, int64_t ldo, int Hp, int Wp, int p) {
    const int h = Hp / p, w = Wp / p;
    const int q4 = p / 4;                                                // 4-px groups per patch row
    const int64_t total = (int64_t)V * h * w * p * q4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % q4) * 4, py = (int)((i / q4) % p);
        const int64_t m = i / ((int64_t)q4 * p);
        const int pc = (int)(m % w), pr = (int)((m / w) % h), v = (int)(m / ((int64_t)w * h));
        const int y = pr * p + py, x0 = pc * p + px;
        T* dst = out + m * ldo + py * p + px;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = y < H && x0 + q < W;
            const uint8_t* src = img + (((int64_t)v * H + y) * W + x0 + q) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[c * p * p + q] = to_act<T>(in ? norm_px(src[n.to_rgb ? 2 - c : c], c, n) : 0.f);
        }
    }
}

ImgNorm make_norm(const float* mean3, const float* std3, int to_rgb) {
    ImgNorm n;
    for (int c = 0; c < 3; ++c) { n.mean[c] = mean3[c]; n.stdinv[c] = (float)(1.0 / (double)std3[c]); }
    n.to_rgb = to_rgb;
    return n;
}

}  // namespace

extern "C" {

int toc3d_normalize_images(const uint8_t* img, int64_t V, int64_t H, int64_t W, const float* mean3, const float* std3, int to_rgb,
                           float* out, int64_t Hp, int64_t Wp, toc3d_stream_t stream) {
    TOC3D_REQUIRE(img && out && mean3 && std3, "toc3d_normalize_images: null buffer");
    TOC3D_REQUIRE(V > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W && Wp % 4 == 0, "toc3d_normalize_images: bad dims (Wp must be a multiple of 4)");
    TOC3D_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "toc3d_normalize_images: zero std");
    const int64_t total = V * Hp * (Wp / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    toc3d_launch(normalize_images_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), img, (int)V, (int)H, (int)W, make_norm(mean3, std3, to_rgb),
                       out, (int)Hp, (int)Wp);
    TOC3D_LAUNCH_CHECK("toc3d_normalize_images");
    return TOC3D_OK;
}

int toc3d_im2col_patches_u8(int dtype, const uint8_t* img, int64_t V, int64_t H, int64_t W, const float* mean3, const float* std3, int to_rgb,
                            void* out, int64_t ldo, int64_t Hp, int64_t Wp, int64_t patch, toc3d_stream_t stream) {
    TOC3D_REQUIRE(img && out && mean3 && std3, "toc3d_im2col_patches_u8: null buffer");
    TOC3D_REQUIRE(V > 0 && H > 0 && W > 0 && patch > 0 && patch % 4 == 0 && Hp >= H && Wp >= W && Hp % patch == 0 && Wp % patch == 0,
                  "toc3d_im2col_patches_u8: bad dims (padded size must be a multiple of the patch, patch a multiple of 4)");
    TOC3D_REQUIRE(ldo >= 3 * patch * patch, "toc3d_im2col_patches_u8: ldo < 3*patch*patch");
    TOC3D_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "toc3d_im2col_patches_u8: zero std");
    const int64_t total = V * (Hp / patch) * (Wp / patch) * patch * (patch / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    const ImgNorm n = make_norm(mean3, std3, to_rgb);
    if (dtype == TOC3D_BF16)
        toc3d_launch(im2col_u8_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), img, (int)V, (int)H, (int)W, n, (bf16_t*)out, ldo, (int)Hp, (int)Wp, (int)patch);
    else if (dtype == TOC3D_F32)
        toc3d_launch(im2col_u8_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), img, (int)V, (int)H, (int)W, n, (float*)out, ldo, (int)Hp, (int)Wp, (int)patch);
    else { toc3d_set_error("toc3d_im2col_patches_u8: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_im2col_patches_u8");
    return TOC3D_OK;
}

// argument checks of toc3d_linear_fused / of one op of toc3d_linear_chain; fills the kernel's argument block (M == 0 is accepted: a.M = 0)
static int fused_args(GemmArgs& a, int dtype, int epilogue, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                      void* out, int64_t ldo, const float* residual, int64_t ldr, int64_t residual_row_mod,
                      float* rep_out, const int32_t* rep_index, int64_t M, int64_t N, int64_t K, int64_t n_valid,
                      float* stats_out, int64_t stats_out_cap, const float* stats_in, int64_t stats_in_cap, const float* col_sums, int64_t ln_n, float ln_eps,
                      void* out_act, int64_t ld_act, const int32_t* residual_index) {
    TOC3D_REQUIRE(dtype == TOC3D_F32 || dtype == TOC3D_BF16 || dtype == TOC3D_F32X3 || dtype == TOC3D_F32X6 || (dtype >= TOC3D_F32X3W && dtype <= TOC3D_F32X3WA),
                  "toc3d_linear: bad dtype %d", dtype);
    // bf16 x 3 on (hi, lo) planes: W always; A (F32X3P, F32X3WA); the outputs a later GEMM multiplies (F32X3P, F32X3WO)
    const bool planes = dtype >= TOC3D_F32X3W && dtype <= TOC3D_F32X3WA;
    const bool planes_a = dtype == TOC3D_F32X3P || dtype == TOC3D_F32X3WA, planes_o = dtype == TOC3D_F32X3P || dtype == TOC3D_F32X3WO;
    if (planes) dtype = TOC3D_F32X3;
    const bool x3_fold = dtype == TOC3D_F32X3 && (epilogue == TOC3D_EPI_SWIGLU_STATS || epilogue == TOC3D_EPI_RESIDUAL_LN || epilogue == TOC3D_EPI_RESIDUAL_STATS ||
                                                  epilogue == TOC3D_EPI_SWIGLU_STATS_LN);
    TOC3D_REQUIRE((dtype != TOC3D_F32X3 && dtype != TOC3D_F32X6) || epilogue <= TOC3D_EPI_GELU || epilogue == TOC3D_EPI_CONV3X3 || x3_fold,
                  "toc3d_linear: the bf16 x 3 / x 6 product forms serve epilogues 0-3 and the 3x3 conv (x 3 also the folded LayerNorms, epilogues 4-7)");
    TOC3D_REQUIRE(A && W && out, "toc3d_linear: null buffer");
    TOC3D_REQUIRE(M >= 0 && N > 0 && K > 0, "toc3d_linear: bad dims M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    const int bk = 64;
    TOC3D_REQUIRE(K % bk == 0, "toc3d_linear: K=%lld must be a multiple of %d (pad at pack time)", (long long)K, bk);
    TOC3D_REQUIRE((lda >= K || epilogue == TOC3D_EPI_CONV3X3) && ldw >= K, "toc3d_linear: leading dims smaller than K");
    TOC3D_REQUIRE((lda * (dtype == TOC3D_BF16 ? 2 : 4)) % 16 == 0 && (ldw * (dtype == TOC3D_BF16 ? 2 : 4)) % 16 == 0,
                  "toc3d_linear: rows must be 16-byte aligned");
    TOC3D_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "toc3d_linear: A/W must be 16-byte aligned");
    const bool e_stats_out = epilogue == TOC3D_EPI_SWIGLU_STATS || epilogue == TOC3D_EPI_SWIGLU_STATS_LN || epilogue == TOC3D_EPI_RESIDUAL_STATS;
    const bool e_ln_in = epilogue == TOC3D_EPI_RESIDUAL_LN || epilogue == TOC3D_EPI_SWIGLU_STATS_LN;
    const bool e_swiglu = epilogue == TOC3D_EPI_SWIGLU || epilogue == TOC3D_EPI_SWIGLU_STATS || epilogue == TOC3D_EPI_SWIGLU_STATS_LN;
    const bool e_residual = epilogue == TOC3D_EPI_RESIDUAL || epilogue == TOC3D_EPI_RESIDUAL_LN || epilogue == TOC3D_EPI_RESIDUAL_STATS;
    TOC3D_REQUIRE(epilogue >= 0 && epilogue <= TOC3D_EPI_CONV3X3, "toc3d_linear: epilogue %d is not served by this entry point", epilogue);
    if (epilogue >= TOC3D_EPI_SWIGLU_STATS && epilogue != TOC3D_EPI_CONV3X3) TOC3D_REQUIRE(dtype == TOC3D_BF16 || x3_fold, "toc3d_linear: the folded-LayerNorm epilogues are bf16 only (and bf16 x 3 on f32 buffers)");
    if (e_stats_out) {
        TOC3D_REQUIRE(stats_out && ((uintptr_t)stats_out % 16) == 0, "toc3d_linear: epilogue %d needs a 16-byte aligned stats_out buffer", epilogue);
        const int64_t slot = e_swiglu ? 128 : 64;
        TOC3D_REQUIRE(stats_out_cap >= (N + slot - 1) / slot, "toc3d_linear: stats_out_cap %lld < ceil(N / %lld)", (long long)stats_out_cap, (long long)slot);
    }
    if (e_ln_in) {
        TOC3D_REQUIRE(stats_in && ((uintptr_t)stats_in % 16) == 0 && (stats_in_cap & 0xffffffff) > 0 && (stats_in_cap >> 32) <= (stats_in_cap & 0xffffffff),
                      "toc3d_linear: epilogue %d needs a 16-byte aligned stats_in buffer (and at most stats_in_cap slots per row)", epilogue);
        TOC3D_REQUIRE(bias && col_sums && ln_n > 0, "toc3d_linear: epilogue %d needs bias (c2), col_sums (c1) and ln_n", epilogue);
        TOC3D_REQUIRE(stats_in != stats_out, "toc3d_linear: stats_in and stats_out must be different buffers");
    }
    if (epilogue == TOC3D_EPI_RESIDUAL_STATS)
        TOC3D_REQUIRE(out_act && ld_act >= N && ((uintptr_t)out_act % 8) == 0 && ld_act % 4 == 0, "toc3d_linear: EPI_RESIDUAL_STATS needs out_act [M, ld_act >= N], 8-byte aligned rows");
    if (e_swiglu) {
        TOC3D_REQUIRE(bias && N % 32 == 0 && n_valid > 0 && n_valid <= N / 2, "toc3d_linear: swiglu needs bias, N%%32==0, n_valid");
        TOC3D_REQUIRE(ldo >= N / 2, "toc3d_linear: swiglu ldo < N/2");
    } else {
        TOC3D_REQUIRE(ldo >= N, "toc3d_linear: ldo < N");
    }
    TOC3D_REQUIRE(!residual_index || (e_residual && residual && residual_row_mod == 0), "toc3d_linear: residual_index needs a residual epilogue, a residual buffer and no row modulus");
    if (e_residual) {
        TOC3D_REQUIRE(!residual || ldr >= N, "toc3d_linear: ldr < N");
        TOC3D_REQUIRE(!rep_index || rep_out, "toc3d_linear: rep_index set without rep_out");
    }
    // 4-wide epilogue accesses: every row start and column group must be 16-byte aligned in its own element size
    const int64_t osz = e_residual ? 4 : (dtype == TOC3D_BF16 ? 2 : 4);
    const bool vec = ldo % 4 == 0 && (uintptr_t)out % (4 * osz) == 0 && (!residual || (ldr % 4 == 0 && (uintptr_t)residual % 16 == 0)) &&
                     (!rep_out || (N % 4 == 0 && (uintptr_t)rep_out % 16 == 0));
    // wide (16-byte) bf16 stores need 16-byte aligned rows in the output's own element size (TOC3D_WIDE_STORES=0 disables them: A/B runs)
    static const bool wide_ok = [] { const char* e = getenv("TOC3D_WIDE_STORES"); return !(e && e[0] == '0'); }();
    const bool vec8 = wide_ok && vec && dtype == TOC3D_BF16 && !e_residual && epilogue != TOC3D_EPI_CONV3X3 && ldo % 8 == 0 && (uintptr_t)out % 16 == 0;
    a = GemmArgs{A, lda, W, ldw, bias, out, ldo, residual, ldr, (int)residual_row_mod, residual_index, rep_out, rep_index,
               (int)M, (int)N, (int)K, (int)n_valid, 0, vec ? 1 : 0, vec8 ? 1 : 0,
               stats_out, (int)stats_out_cap, stats_in, (int)(stats_in_cap & 0xffffffff), (int)(stats_in_cap >> 32), col_sums, ln_n > 0 ? (float)(1.0 / (double)ln_n) : 0.f, ln_eps, out_act, ld_act,
               0, 0, nullptr, nullptr, nullptr, 0, 1.0f, planes_a, planes, planes_o, 0, nullptr, nullptr, dtype == TOC3D_F32X3 ? 1 : 0};
    if (planes) {
        TOC3D_REQUIRE(epilogue != TOC3D_EPI_CONV3X3 || !planes_a, "toc3d_linear: the 3x3 conv gathers f32 activations (A cannot be planes)");
        const void* act_copy = epilogue == TOC3D_EPI_CONV3X3 ? nullptr : out_act;     // (the conv carries its zero line and h << 32 | w in these two slots)
        TOC3D_REQUIRE((!planes_o || ((!e_swiglu || (ldo % 32 == 0 && (uintptr_t)out % 16 == 0)) && (!act_copy || (ld_act % 32 == 0 && (uintptr_t)act_copy % 16 == 0)))) &&
                      (!planes_a || lda % 32 == 0),
                      "toc3d_linear: rows of (hi, lo) planes are whole 32-element groups on 16-byte boundaries: lda, ldo (SwiGLU) and ld_act must be multiples of 32");
    }
    if (epilogue == TOC3D_EPI_CONV3X3) {
        // A = NHWC act tensor [V, h, w, lda]; ld_act carries h << 32 | w and out_act the zero line (toc3d_conv3x3_nhwc fills them in)
        a.conv_h = (int)(ld_act >> 32); a.conv_w = (int)(ld_act & 0xffffffff); a.zeros = out_act;
        TOC3D_REQUIRE(a.conv_h > 0 && a.conv_w > 0 && a.zeros && M % ((int64_t)a.conv_h * a.conv_w) == 0 && K == 9 * lda, "toc3d_linear: EPI_CONV3X3 goes through toc3d_conv3x3_nhwc");
        a.out_act = nullptr; a.ld_act = 0;
    }
    return TOC3D_OK;
}

// Deterministic split-K (variant = 1000 * split + tile variant): workspace = [TOC3D_SPLITK_TICKET_BYTES of arrival tickets | f32 partial tiles].  The ticket
// region has a FIXED size, so launches of different shapes may share one workspace on one stream (the partials are scratch between launches, the tickets are
// zero between launches: every tile's last arriver re-arms its word).
static_assert(TOC3D_SPLITK_TICKET_BYTES % 256 == 0, "the partial tiles start on a 256-byte boundary");
int64_t toc3d_linear_splitk_workspace_bytes(int variant, int64_t M, int64_t N) {
    const int split = variant / 1000;
    const int dims = toc3d_gemm_splitk_tile_dims(variant % 1000);      // the launch table's own tile shape (the element count alone cannot tell 128x64 from 64x128)
    if (split < 2 || split > TOC3D_SPLITK_MAX || dims == 0 || M < 0 || N <= 0) return -1;
    const int64_t bm = dims >> 16, bn = dims & 0xffff, elems = bm * bn;
    const int64_t tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    if (tiles * 4 > TOC3D_SPLITK_TICKET_BYTES) return -1;
    return TOC3D_SPLITK_TICKET_BYTES + tiles * split * elems * 4;
}

int toc3d_linear_fused_ws(int dtype, int epilogue, int variant, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                          void* out, int64_t ldo, const float* residual, int64_t ldr, int64_t residual_row_mod,
                          float* rep_out, const int32_t* rep_index, int64_t M, int64_t N, int64_t K, int64_t n_valid,
                          float* stats_out, int64_t stats_out_cap, const float* stats_in, int64_t stats_in_cap, const float* col_sums, int64_t ln_n, float ln_eps,
                          void* out_act, int64_t ld_act, const int32_t* residual_index, void* workspace, int64_t workspace_bytes, toc3d_stream_t stream) {
    if (variant < 1000)
        return toc3d_linear_fused(dtype, epilogue, variant, A, lda, W, ldw, bias, out, ldo, residual, ldr, residual_row_mod, rep_out, rep_index, M, N, K, n_valid,
                                  stats_out, stats_out_cap, stats_in, stats_in_cap, col_sums, ln_n, ln_eps, out_act, ld_act, residual_index, stream);
    TOC3D_REQUIRE(epilogue == TOC3D_EPI_RESIDUAL || epilogue == TOC3D_EPI_RESIDUAL_LN || epilogue == TOC3D_EPI_RESIDUAL_STATS,
                  "toc3d_linear_fused_ws: split-K serves the residual epilogues (1, 5, 6), not %d", epilogue);
    const int64_t need = toc3d_linear_splitk_workspace_bytes(variant, M, N);
    TOC3D_REQUIRE(need > 0, "toc3d_linear_fused_ws: variant %d has no split-K form (split 2..%d of tile variants 1, 9, 10, 14, 16, 17, 19, 22, 26, 28, 29, 55, 56), or too many tiles", variant, TOC3D_SPLITK_MAX);
    TOC3D_REQUIRE(workspace && ((uintptr_t)workspace % 256) == 0 && workspace_bytes >= need, "toc3d_linear_fused_ws: workspace of >= %lld bytes, 256-byte aligned (toc3d_linear_splitk_workspace_bytes)", (long long)need);
    TOC3D_REQUIRE(K >= 128 * (variant / 1000), "toc3d_linear_fused_ws: K = %lld is too short for a split of %d", (long long)K, variant / 1000);
    GemmArgs a;
    const int rc_args = fused_args(a, dtype, epilogue, A, lda, W, ldw, bias, out, ldo, residual, ldr, residual_row_mod, rep_out, rep_index, M, N, K, n_valid,
                                   stats_out, stats_out_cap, stats_in, stats_in_cap, col_sums, ln_n, ln_eps, out_act, ld_act, residual_index);
    if (rc_args != TOC3D_OK) return rc_args;
    if (M == 0) return TOC3D_OK;
    a.split = variant / 1000;
    a.sk_tickets = reinterpret_cast<unsigned*>(workspace);
    a.sk_slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + TOC3D_SPLITK_TICKET_BYTES);
    const bool x3 = dtype == TOC3D_F32X3 || (dtype >= TOC3D_F32X3W && dtype <= TOC3D_F32X3WA);
    TOC3D_REQUIRE(dtype == TOC3D_BF16 || dtype == TOC3D_F32 || x3, "toc3d_linear_fused_ws: split-K serves bf16, f32 and the bf16 x 3 forms");
    g_bad_variant = false;
    const int rc = toc3d_gemm_launch_splitk(x3 ? TOC3D_F32X3 : dtype, epilogue, variant % 1000, a, as_stream(stream));
    if (rc != TOC3D_OK) { toc3d_set_error("toc3d_linear_fused_ws: tile variant %d has no split-K form for this dtype", variant % 1000); return rc; }
    if (g_bad_variant) { toc3d_set_error("toc3d_linear_fused_ws: K = %lld is not a whole number of the K-tiles of variant %d", (long long)K, variant % 1000); return TOC3D_ERR_UNSUPPORTED; }
    TOC3D_LAUNCH_CHECK("toc3d_linear_fused_ws");
    return TOC3D_OK;
}

int toc3d_linear_fused(int dtype, int epilogue, int variant, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                       void* out, int64_t ldo, const float* residual, int64_t ldr, int64_t residual_row_mod,
                       float* rep_out, const int32_t* rep_index, int64_t M, int64_t N, int64_t K, int64_t n_valid,
                       float* stats_out, int64_t stats_out_cap, const float* stats_in, int64_t stats_in_cap, const float* col_sums, int64_t ln_n, float ln_eps,
                       void* out_act, int64_t ld_act, const int32_t* residual_index, toc3d_stream_t stream) {
    TOC3D_REQUIRE(variant < 1000, "toc3d_linear: variant %d is a split-K variant: it takes a workspace (toc3d_linear_fused_ws)", variant);
    GemmArgs a;
    const int rc_args = fused_args(a, dtype, epilogue, A, lda, W, ldw, bias, out, ldo, residual, ldr, residual_row_mod, rep_out, rep_index, M, N, K, n_valid,
                                   stats_out, stats_out_cap, stats_in, stats_in_cap, col_sums, ln_n, ln_eps, out_act, ld_act, residual_index);
    if (rc_args != TOC3D_OK) return rc_args;
    if (M == 0) return TOC3D_OK;
    g_bad_variant = false;
    int rc = (dtype == TOC3D_F32X3 || (dtype >= TOC3D_F32X3W && dtype <= TOC3D_F32X3WA)) ? toc3d_gemm_launch_x3(epilogue, variant, a, as_stream(stream))
             : dtype == TOC3D_F32X6 ? toc3d_gemm_launch_x6(epilogue, variant, a, as_stream(stream)) : launch_gemm(dtype == TOC3D_BF16, epilogue, variant, a, as_stream(stream));
    if (rc != TOC3D_OK) { toc3d_set_error("toc3d_linear: bad epilogue %d or variant %d", epilogue, variant); return rc; }
    if (g_bad_variant) { toc3d_set_error("toc3d_linear: variant %d cannot serve epilogue %d (per-wave column slab not a multiple of 32, or N-tile not a multiple of 128 for the statistics)", variant, epilogue); return TOC3D_ERR_UNSUPPORTED; }
    TOC3D_LAUNCH_CHECK("toc3d_linear");
    return TOC3D_OK;
}


int toc3d_conv3x3_nhwc(int dtype, int variant, const void* x, int64_t C, const void* W, int64_t ldw, const float* bias, float* out, int64_t ldo,
                       int64_t V, int64_t h, int64_t w, int64_t Cout, const void* zeros, toc3d_stream_t stream) {
    TOC3D_REQUIRE(x && W && out && zeros && V > 0 && h > 0 && w > 0 && h < 32768 && w < 65536, "toc3d_conv3x3_nhwc: bad arguments");
    TOC3D_REQUIRE(C % 64 == 0, "toc3d_conv3x3_nhwc: the channel count must be a multiple of 64 (one K-tile never straddles two taps)");
    return toc3d_linear_fused(dtype, TOC3D_EPI_CONV3X3, variant, x, C, W, ldw, bias, out, ldo, nullptr, 0, 0, nullptr, nullptr, V * h * w, Cout, 9 * C, 0,
                              nullptr, 0, nullptr, 0, nullptr, 0, 0.f, const_cast<void*>(zeros), (h << 32) | w, nullptr, stream);
}

int toc3d_linear_qkv_rope(int dtype, int variant, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo,
                          int64_t M, int64_t N, int64_t K, const int32_t* rope_rc, const float* rope_tab, int64_t rope_side,
                          float q_scale, toc3d_stream_t stream) {
    // bf16, or bf16 x 3 on (hi, lo) planes (round 6): W in planes, the rotated q|k|v rows written as planes; A in planes (F32X3P) or plain f32 (F32X3WO)
    const bool x3 = dtype == TOC3D_F32X3P || dtype == TOC3D_F32X3WO;
    TOC3D_REQUIRE(dtype == TOC3D_BF16 || x3, "toc3d_linear_qkv_rope: bf16, TOC3D_DTYPE_F32X3P or F32X3WO (exact f32 rotates inside toc3d_window_attention)");
    TOC3D_REQUIRE(A && W && out && rope_rc && rope_tab, "toc3d_linear_qkv_rope: null buffer");
    TOC3D_REQUIRE(M >= 0 && N > 0 && N % 192 == 0 && K > 0 && K % 64 == 0, "toc3d_linear_qkv_rope: N = 3C with C a multiple of 64, K a multiple of 64");
    const int esz = x3 ? 4 : 2;
    TOC3D_REQUIRE(lda >= K && ldw >= K && ldo >= N && (lda * esz) % 16 == 0 && (ldw * esz) % 16 == 0, "toc3d_linear_qkv_rope: bad leading dims");
    TOC3D_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)rope_tab % 16) == 0, "toc3d_linear_qkv_rope: misaligned buffer");
    TOC3D_REQUIRE(rope_side > 0 && rope_side <= 64, "toc3d_linear_qkv_rope: rope_side out of range (the tables live in LDS: <= 64; launch_cfg / launch_phased size their LDS attribute for 64)");
    if (x3) TOC3D_REQUIRE(ldo % 32 == 0 && ((uintptr_t)out % 128) == 0 && (dtype != TOC3D_F32X3P || lda % 32 == 0), "toc3d_linear_qkv_rope: rows of (hi, lo) planes are whole 32-element groups on 128-byte boundaries");
    if (M == 0) return TOC3D_OK;
    const bool vec = ldo % 4 == 0 && (uintptr_t)out % (4 * esz) == 0;
    static const bool wide_ok = [] { const char* e = getenv("TOC3D_WIDE_STORES"); return !(e && e[0] == '0'); }();
    const bool vec8 = !x3 && wide_ok && vec && ldo % 8 == 0 && (uintptr_t)out % 16 == 0;
    GemmArgs a{A, lda, W, ldw, bias, out, ldo, nullptr, 0, 0, nullptr, nullptr, nullptr, (int)M, (int)N, (int)K, 0, 0, vec ? 1 : 0, vec8 ? 1 : 0,
               nullptr, 0, nullptr, 0, 0, nullptr, 0.f, 0.f, nullptr, 0, 0, 0, nullptr, rope_rc, rope_tab, (int)rope_side, q_scale,
               dtype == TOC3D_F32X3P ? 1 : 0, x3 ? 1 : 0, x3 ? 1 : 0};
    g_bad_variant = false;
    const int rc = x3 ? toc3d_gemm_launch_x3(TOC3D_EPI_QKV_ROPE, variant, a, as_stream(stream)) : launch_gemm(1, TOC3D_EPI_QKV_ROPE, variant, a, as_stream(stream));
    if (rc != TOC3D_OK) { toc3d_set_error("toc3d_linear_qkv_rope: bad variant %d", variant); return rc; }
    if (g_bad_variant) { toc3d_set_error("toc3d_linear_qkv_rope: variant %d cannot serve this epilogue", variant); return TOC3D_ERR_UNSUPPORTED; }
    TOC3D_LAUNCH_CHECK("toc3d_linear_qkv_rope");
    return TOC3D_OK;
}


int toc3d_linear_ex(int dtype, int epilogue, int variant, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                    void* out, int64_t ldo, const float* residual, int64_t ldr, int64_t residual_row_mod,
                    float* rep_out, const int32_t* rep_index, int64_t M, int64_t N, int64_t K, int64_t n_valid,
                    toc3d_stream_t stream) {
    TOC3D_REQUIRE(epilogue < TOC3D_EPI_SWIGLU_STATS, "toc3d_linear_ex: epilogue %d takes the extra arguments of toc3d_linear_fused", epilogue);
    return toc3d_linear_fused(dtype, epilogue, variant, A, lda, W, ldw, bias, out, ldo, residual, ldr, residual_row_mod, rep_out, rep_index,
                              M, N, K, n_valid, nullptr, 0, nullptr, 0, nullptr, 0, 0.f, nullptr, 0, nullptr, stream);
}

int toc3d_linear(int dtype, int epilogue, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                 void* out, int64_t ldo, const float* residual, int64_t ldr, int64_t residual_row_mod,
                 float* rep_out, const int32_t* rep_index, int64_t M, int64_t N, int64_t K, int64_t n_valid,
                 toc3d_stream_t stream) {
    return toc3d_linear_ex(dtype, epilogue, 0, A, lda, W, ldw, bias, out, ldo, residual, ldr, residual_row_mod, rep_out, rep_index,
                           M, N, K, n_valid, stream);
}

int toc3d_x3_planes(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int64_t K, toc3d_stream_t stream) {
    TOC3D_REQUIRE(src && dst && rows >= 0 && K > 0 && K % 32 == 0 && ld_src >= K && ld_dst >= K && ld_dst % 32 == 0, "toc3d_x3_planes: K and ld_dst are multiples of 32, leading dims >= K");
    TOC3D_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0 && ld_src % 4 == 0, "toc3d_x3_planes: 16-byte aligned rows");
    TOC3D_REQUIRE(src != dst || ld_src == ld_dst, "toc3d_x3_planes: in place needs equal leading dims");
    if (rows == 0) return TOC3D_OK;
    const int64_t groups = rows * (K / 32);
    toc3d_launch(x3_planes_kernel, dim3((unsigned)((groups + 31) / 32)), dim3(256), 0, as_stream(stream), src, ld_src, dst, ld_dst, rows, (int)(K / 32));
    TOC3D_LAUNCH_CHECK("toc3d_x3_planes");
    return TOC3D_OK;
}

int toc3d_pack_weight(int dtype, const float* w, int64_t N, int64_t K, void* out, int64_t Np, int64_t Kp, toc3d_stream_t stream) {
    TOC3D_REQUIRE(w && out && Np >= N && Kp >= K && N > 0 && K > 0, "toc3d_pack_weight: bad arguments");
    const int blocks = (int)((Np * Kp + 255) / 256 < 4096 ? (Np * Kp + 255) / 256 : 4096);
    if (dtype == TOC3D_BF16)
        toc3d_launch(pack_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), w, (int)N, (int)K, (bf16_t*)out, (int)Np, (int)Kp);
    else if (dtype == TOC3D_F32)
        toc3d_launch(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), w, (int)N, (int)K, (float*)out, (int)Np, (int)Kp);
    else { toc3d_set_error("toc3d_pack_weight: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_pack_weight");
    return TOC3D_OK;
}

int toc3d_pack_weight_lnfold(int dtype, const float* w3, const float* gamma, const float* beta, const float* b3, int64_t N, int64_t K,
                             void* out_w, int64_t Np, int64_t Kp, float* c1, float* c2, toc3d_stream_t stream) {
    TOC3D_REQUIRE(w3 && gamma && beta && b3 && out_w && c1 && c2 && Np >= N && Kp >= K && N > 0 && K > 0, "toc3d_pack_weight_lnfold: bad arguments");
    if (dtype == TOC3D_BF16)
        toc3d_launch(pack_lnfold_kernel<bf16_t>, dim3((unsigned)Np), dim3(256), 0, as_stream(stream), w3, gamma, beta, b3, (int)N, (int)K, (bf16_t*)out_w, (int)Kp, c1, c2);
    else if (dtype == TOC3D_F32)
        toc3d_launch(pack_lnfold_kernel<float>, dim3((unsigned)Np), dim3(256), 0, as_stream(stream), w3, gamma, beta, b3, (int)N, (int)K, (float*)out_w, (int)Kp, c1, c2);
    else { toc3d_set_error("toc3d_pack_weight_lnfold: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_pack_weight_lnfold");
    return TOC3D_OK;
}

int toc3d_pack_swiglu_lnfold(int dtype, const float* w1, const float* w2, const float* b1, const float* b2, const float* gamma, const float* beta,
                             int64_t Hd, int64_t K, void* out_w, float* c1, float* c2, int64_t Hp, int64_t Kp, toc3d_stream_t stream) {
    TOC3D_REQUIRE(w1 && w2 && b1 && b2 && gamma && beta && out_w && c1 && c2, "toc3d_pack_swiglu_lnfold: null buffer");
    TOC3D_REQUIRE(Hp >= Hd && Hp % 64 == 0 && Kp >= K, "toc3d_pack_swiglu_lnfold: Hp must be >= Hd and a multiple of 64");
    if (dtype == TOC3D_BF16)
        toc3d_launch(pack_swiglu_lnfold_kernel<bf16_t>, dim3((unsigned)(2 * Hp)), dim3(256), 0, as_stream(stream), w1, w2, b1, b2, gamma, beta, (int)Hd, (int)K, (bf16_t*)out_w, c1, c2, (int)Kp);
    else if (dtype == TOC3D_F32)
        toc3d_launch(pack_swiglu_lnfold_kernel<float>, dim3((unsigned)(2 * Hp)), dim3(256), 0, as_stream(stream), w1, w2, b1, b2, gamma, beta, (int)Hd, (int)K, (float*)out_w, c1, c2, (int)Kp);
    else { toc3d_set_error("toc3d_pack_swiglu_lnfold: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_pack_swiglu_lnfold");
    return TOC3D_OK;
}

int toc3d_pack_swiglu(int dtype, const float* w1, const float* w2, const float* b1, const float* b2, int64_t Hd, int64_t K,
                      void* out_w, float* out_b, int64_t Hp, int64_t Kp, toc3d_stream_t stream) {
    TOC3D_REQUIRE(w1 && w2 && b1 && b2 && out_w && out_b, "toc3d_pack_swiglu: null buffer");
    TOC3D_REQUIRE(Hp >= Hd && Hp % 64 == 0 && Kp >= K, "toc3d_pack_swiglu: Hp must be >= Hd and a multiple of 64");
    const int64_t total = 2 * Hp * Kp;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == TOC3D_BF16)
        toc3d_launch(pack_swiglu_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), w1, w2, b1, b2, (int)Hd, (int)K, (bf16_t*)out_w, out_b, (int)Hp, (int)Kp);
    else if (dtype == TOC3D_F32)
        toc3d_launch(pack_swiglu_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), w1, w2, b1, b2, (int)Hd, (int)K, (float*)out_w, out_b, (int)Hp, (int)Kp);
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
        toc3d_launch(im2col_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), img, (bf16_t*)out, ldo, (int)V, (int)Cin, (int)H, (int)W, (int)patch);
    else if (dtype == TOC3D_F32)
        toc3d_launch(im2col_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), img, (float*)out, ldo, (int)V, (int)Cin, (int)H, (int)W, (int)patch);
    else { toc3d_set_error("toc3d_im2col_patches: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_im2col_patches");
    return TOC3D_OK;
}

}  // extern "C"
