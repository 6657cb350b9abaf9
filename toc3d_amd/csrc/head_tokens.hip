// Token side of StreamPETRHead.forward (second half): what consumes the neck's features behind the
// backbone.  Reference: dense_heads/streampetr_head.py position_embeding :378-422, forward :627-639; models/utils/misc.py
// locations :59-82, MLN :154-188, SELayer_Linear :139-151.  The linear layers run on toc3d_linear; this file holds the elementwise /
// geometry kernels around them.  All HBM-bound byte work on [tokens, 64..1024] rows.
#include "capi.h"
#include "common.h"

namespace {

struct FrustumArgs {
    const float* img2lidar;     // [B*N, 16] inverse of lidar2img
    const float* intrinsics;    // [B*N, 16]
    const float* coords_d;      // [D]
    float pr[6];                // position_range
    int B, N, h, w, D, stride, pad_h, pad_w;
};

TOC3D_DEV float inv_sigmoid(float x) {                                  // mmdet inverse_sigmoid, eps = 1e-5
    x = fminf(fmaxf(x, 0.f), 1.f);
    return logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
}

// one thread per (token, depth bin): the 3-D point of the pixel centre at that depth through lidar2img^-1 (:391-413)
template <typename T>
__global__ __launch_bounds__(256) void frustum_kernel(FrustumArgs a, T* __restrict__ pin, int64_t ld_pin, T* __restrict__ cone_act,
                                                      int64_t ld_cone, float* __restrict__ cone) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int hw = a.h * a.w;
    const int64_t total = (int64_t)a.B * a.N * hw * a.D;
    if (id >= total) return;
    const int d = (int)(id % a.D);
    const int64_t tokg = id / a.D;                                       // b * N*hw + n*hw + y*w + x
    const int pix = (int)(tokg % hw), view = (int)(tokg / hw);           // view = b*N + n
    const int y = pix / a.w, x = pix % a.w;
    // locations() (misc.py:70-77) then * pad (streampetr_head.py:392-393), as the reference rounds them
    const float cx = __fmul_rn(__fdiv_rn((float)(x * a.stride + a.stride / 2), (float)a.pad_w), (float)a.pad_w);
    const float cy = __fmul_rn(__fdiv_rn((float)(y * a.stride + a.stride / 2), (float)a.pad_h), (float)a.pad_h);
    const float dep = a.coords_d[d];
    const float sc = fmaxf(dep, 1e-5f);
    const float c0 = __fmul_rn(cx, sc), c1 = __fmul_rn(cy, sc);
    const float* M = a.img2lidar + (int64_t)view * 16;
    float p[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float s = __fmul_rn(M[i * 4], c0);
        s = __fadd_rn(s, __fmul_rn(M[i * 4 + 1], c1));
        s = __fadd_rn(s, __fmul_rn(M[i * 4 + 2], dep));
        s = __fadd_rn(s, M[i * 4 + 3]);
        p[i] = __fdiv_rn(__fsub_rn(s, a.pr[i]), __fsub_rn(a.pr[3 + i], a.pr[i]));
    }
    T* dst = pin + tokg * ld_pin + d * 3;
    dst[0] = to_act<T>(inv_sigmoid(p[0]));
    dst[1] = to_act<T>(inv_sigmoid(p[1]));
    dst[2] = to_act<T>(inv_sigmoid(p[2]));
    // cone (:419-420): [|fx|, |fy|] / 1e3 of camera (token index % N) -- the reference's repeat order (:385-386) --, then the points
    // of the last depth bin and of bin D - 30
    float* cr = cone + tokg * 8;
    T* ca = cone_act + tokg * ld_cone;
    if (d == a.D - 1 || d == a.D - 30) {
        const int o = d == a.D - 1 ? 2 : 5;
#pragma unroll
        for (int i = 0; i < 3; ++i) { cr[o + i] = p[i]; ca[o + i] = to_act<T>(p[i]); }
    }
    if (d == 0) {
        const int b = view / a.N;
        const int tok_in_b = (int)(tokg - (int64_t)b * a.N * hw);
        const float* K = a.intrinsics + (int64_t)(b * a.N + tok_in_b % a.N) * 16;
        const float fx = __fdiv_rn(fabsf(K[0]), 1e3f), fy = __fdiv_rn(fabsf(K[5]), 1e3f);
        cr[0] = fx; cr[1] = fy;
        ca[0] = to_act<T>(fx); ca[1] = to_act<T>(fy);
    }
}

template <typename T>
__global__ void relu_kernel(T* __restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = from_act(x[i]); x[i] = to_act<T>(v > 0.f ? v : 0.f); }
}

// NCHW f32 [V, C, hw] -> rows [V*hw, ld] act
template <typename T>
__global__ __launch_bounds__(256) void nchw_rows_kernel(const float* __restrict__ x, T* __restrict__ out, int64_t ld, int C, int hw) {
    __shared__ float tile[32][33];
    const int v = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        tile[r][tx] = (c < C && t < hw) ? x[((int64_t)v * C + c) * hw + t] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        if (t < hw && c < C) out[((int64_t)v * hw + t) * ld + c] = to_act<T>(tile[tx][r]);
    }
}

// MLN.forward (misc.py:181-188): out = gamma * LayerNorm(x; no affine, eps 1e-5) + beta; one wavefront per row, E <= 1024
template <typename T>
__global__ __launch_bounds__(256) void mln_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  int E, float* __restrict__ out, T* __restrict__ out_act, int64_t ld_act, int M) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; v[i] = c < E ? x[(int64_t)row * E + c] : 0.f; s += v[i]; }
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; const float dlt = c < E ? v[i] - mean : 0.f; q += dlt * dlt; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + 1e-5f);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < E) {
            const float o = gamma[(int64_t)row * E + c] * ((v[i] - mean) * rstd) + beta[(int64_t)row * E + c];
            out[(int64_t)row * E + c] = o;
            out_act[(int64_t)row * ld_act + c] = to_act<T>(o);
        }
    }
}

// SELayer_Linear gate (misc.py:151): out = pos * sigmoid(se)
__global__ void se_gate_kernel(const float* __restrict__ pos, const float* __restrict__ se, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pos[i] * (1.0f / (1.0f + expf(-se[i])));
}

}  // namespace

extern "C" {

int toc3d_head_frustum_inputs(int dtype, const float* img2lidar, const float* intrinsics, const float* coords_d, const float* position_range,
                              int64_t B, int64_t N, int64_t h, int64_t w, int64_t D, int64_t stride, int64_t pad_h, int64_t pad_w,
                              void* pos_in, int64_t ld_pos, void* cone_act, int64_t ld_cone, float* cone, toc3d_stream_t stream) {
    TOC3D_REQUIRE(img2lidar && intrinsics && coords_d && position_range && pos_in && cone_act && cone, "toc3d_head_frustum_inputs: null buffer");
    TOC3D_REQUIRE(B > 0 && N > 0 && h > 0 && w > 0 && D >= 30 && stride > 0 && pad_h > 0 && pad_w > 0 && ld_pos >= 3 * D && ld_cone >= 8,
                  "toc3d_head_frustum_inputs: bad dims (depth_num >= 30, ld_pos >= 3*depth_num, ld_cone >= 8)");
    FrustumArgs a;
    a.img2lidar = img2lidar; a.intrinsics = intrinsics; a.coords_d = coords_d;
    for (int i = 0; i < 6; ++i) a.pr[i] = position_range[i];              // host pointer
    a.B = (int)B; a.N = (int)N; a.h = (int)h; a.w = (int)w; a.D = (int)D; a.stride = (int)stride; a.pad_h = (int)pad_h; a.pad_w = (int)pad_w;
    const int64_t total = B * N * h * w * D;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == TOC3D_BF16) toc3d_launch(frustum_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), a, (bf16_t*)pos_in, ld_pos, (bf16_t*)cone_act, ld_cone, cone);
    else if (dtype == TOC3D_F32) toc3d_launch(frustum_kernel<float>, grid, dim3(256), 0, as_stream(stream), a, (float*)pos_in, ld_pos, (float*)cone_act, ld_cone, cone);
    else { toc3d_set_error("toc3d_head_frustum_inputs: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_head_frustum_inputs");
    return TOC3D_OK;
}

int toc3d_relu_inplace(int dtype, void* x, int64_t n, toc3d_stream_t stream) {
    TOC3D_REQUIRE(x && n >= 0, "toc3d_relu_inplace: bad arguments");
    if (n == 0) return TOC3D_OK;
    dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == TOC3D_BF16) toc3d_launch(relu_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (bf16_t*)x, n);
    else if (dtype == TOC3D_F32) toc3d_launch(relu_kernel<float>, grid, dim3(256), 0, as_stream(stream), (float*)x, n);
    else { toc3d_set_error("toc3d_relu_inplace: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_relu_inplace");
    return TOC3D_OK;
}

int toc3d_nchw_to_rows(int dtype, const float* x, void* out, int64_t ldo, int64_t V, int64_t C, int64_t hw, toc3d_stream_t stream) {
    TOC3D_REQUIRE(x && out && V > 0 && C > 0 && hw > 0 && ldo >= C && V <= 65535, "toc3d_nchw_to_rows: bad arguments");
    dim3 grid((unsigned)((hw + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)V);
    if (dtype == TOC3D_BF16) toc3d_launch(nchw_rows_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), x, (bf16_t*)out, ldo, (int)C, (int)hw);
    else if (dtype == TOC3D_F32) toc3d_launch(nchw_rows_kernel<float>, grid, dim3(256), 0, as_stream(stream), x, (float*)out, ldo, (int)C, (int)hw);
    else { toc3d_set_error("toc3d_nchw_to_rows: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_nchw_to_rows");
    return TOC3D_OK;
}

int toc3d_mln_apply(int dtype, const float* x, const float* gamma, const float* beta, int64_t M, int64_t E, float* out, void* out_act, int64_t ld_act,
                    toc3d_stream_t stream) {
    TOC3D_REQUIRE(x && gamma && beta && out && out_act && M >= 0 && E > 0 && E <= 1024 && ld_act >= E, "toc3d_mln_apply: bad arguments (E <= 1024)");
    if (M == 0) return TOC3D_OK;
    dim3 grid((unsigned)((M + 3) / 4));
    if (dtype == TOC3D_BF16) toc3d_launch(mln_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), x, gamma, beta, (int)E, out, (bf16_t*)out_act, ld_act, (int)M);
    else if (dtype == TOC3D_F32) toc3d_launch(mln_kernel<float>, grid, dim3(256), 0, as_stream(stream), x, gamma, beta, (int)E, out, (float*)out_act, ld_act, (int)M);
    else { toc3d_set_error("toc3d_mln_apply: bad dtype"); return TOC3D_ERR_ARG; }
    TOC3D_LAUNCH_CHECK("toc3d_mln_apply");
    return TOC3D_OK;
}

int toc3d_se_gate(const float* pos, const float* se, float* out, int64_t n, toc3d_stream_t stream) {
    TOC3D_REQUIRE(pos && se && out && n >= 0, "toc3d_se_gate: bad arguments");
    if (n == 0) return TOC3D_OK;
    toc3d_launch(se_gate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), pos, se, out, n);
    TOC3D_LAUNCH_CHECK("toc3d_se_gate");
    return TOC3D_OK;
}

}  // extern "C"
