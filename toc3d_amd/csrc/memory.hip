// StreamPETR temporal memory bank on the GPU: the producer of the backbone scorer's inputs.
// Reference: dense_heads/streampetr_head.py  pre_update_memory :322-346,  post_update_memory :348-377; helpers
// models/utils/misc.py memory_refresh :7-11, topk_gather :13-23, transform_reference_points.
// Byte / index work on small tensors (B x 640 x 256): one thread per (sample, slot, column); HBM/latency bound, no LDS.
#include "capi.h"
#include "common.h"

namespace {

// 4x4 row-major pose times 4x4 / homogeneous point; plain mul + add in k order (no contraction: -ffp-contract would fuse
// differently per call site)
TOC3D_DEV float dot4(const float* a, float b0, float b1, float b2, float b3) {
    float s = __fmul_rn(a[0], b0);
    s = __fadd_rn(s, __fmul_rn(a[1], b1));
    s = __fadd_rn(s, __fmul_rn(a[2], b2));
    s = __fadd_rn(s, __fmul_rn(a[3], b3));
    return s;
}

struct Bank { float* emb; float* ref; double* ts; float* pose; float* vel; };

// In place on slots [0, memory_len).  fresh != 0: the bank was just zero-filled (reset_memory + first pre_update, :326-331).
__global__ void pre_update_kernel(Bank m, const float* __restrict__ prev_exists, const double* __restrict__ timestamp,
                                  const float* __restrict__ pose_inv, const float* __restrict__ pseudo, const float* __restrict__ pc_range,
                                  int B, int cap, int memory_len, int num_propagated, int D, int fresh) {
    const int l = blockIdx.x, b = blockIdx.y;
    const float x = prev_exists[b];
    const int64_t s = (int64_t)b * cap + l;
    const float* P = pose_inv + b * 16;
    if (!fresh) {
        for (int d = threadIdx.x; d < D; d += blockDim.x) m.emb[s * D + d] = __fmul_rn(m.emb[s * D + d], x);
        if (threadIdx.x < 2) m.vel[s * 2 + threadIdx.x] = __fmul_rn(m.vel[s * 2 + threadIdx.x], x);
        if (threadIdx.x == 2) m.ts[s] = (m.ts[s] + timestamp[b]) * (double)x;                       // :333,336
        __shared__ float old_pose[16], old_ref[3];
        if (threadIdx.x < 16) old_pose[threadIdx.x] = m.pose[s * 16 + threadIdx.x];
        if (threadIdx.x >= 32 && threadIdx.x < 35) old_ref[threadIdx.x - 32] = m.ref[s * 3 + threadIdx.x - 32];
        __syncthreads();
        if (threadIdx.x < 16) {                                                                     // :334 pose_inv @ pose
            const int i = threadIdx.x >> 2, j = threadIdx.x & 3;
            m.pose[s * 16 + threadIdx.x] = __fmul_rn(dot4(P + i * 4, old_pose[j], old_pose[4 + j], old_pose[8 + j], old_pose[12 + j]), x);
        }
        if (threadIdx.x >= 32 && threadIdx.x < 35) {                                                // :335 transform_reference_points
            const int i = threadIdx.x - 32;
            m.ref[s * 3 + i] = __fmul_rn(dot4(P + i * 4, old_ref[0], old_ref[1], old_ref[2], 1.0f), x);
        }
        __syncthreads();
    }
    if (l < num_propagated) {                                                                       // :343-346
        const float nx = __fsub_rn(1.0f, x);
        if (threadIdx.x < 3) {
            const int i = threadIdx.x;
            const float ps = __fadd_rn(__fmul_rn(pseudo[l * 3 + i], __fsub_rn(pc_range[3 + i], pc_range[i])), pc_range[i]);
            m.ref[s * 3 + i] = __fadd_rn(m.ref[s * 3 + i], __fmul_rn(nx, ps));
        }
        if (threadIdx.x >= 32 && threadIdx.x < 36) {
            const int i = threadIdx.x - 32;
            m.pose[s * 16 + i * 5] = __fadd_rn(m.pose[s * 16 + i * 5], nx);
        }
    }
}

// rec_score = sigmoid(cls).topk(1).values (:361): max over classes of the f32 sigmoid
__global__ void score_kernel(const float* __restrict__ cls, int64_t n, int ncls, float* __restrict__ score) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float best = -1.f;
    for (int c = 0; c < ncls; ++c) {
        const float sg = 1.0f / (1.0f + expf(-cls[i * ncls + c]));
        best = fmaxf(best, sg);
    }
    score[i] = best;
}

// out = transform(cat([gather(rec, top-k), in[:memory_len]]))   (:365-377); in and out are distinct banks
__global__ void post_update_kernel(Bank in, Bank out, const int64_t* __restrict__ order, const float* __restrict__ rec_pose,
                                   const float* __restrict__ bbox, int ld_bbox, const float* __restrict__ dec, const float* __restrict__ ego_pose,
                                   const double* __restrict__ timestamp, int B, int Q, int cap, int topk, int D) {
    const int l = blockIdx.x, b = blockIdx.y;
    const int64_t so = (int64_t)b * cap + l;
    const float* E = ego_pose + b * 16;
    __shared__ float src_pose[16], src_ref[3];
    const bool fresh = l < topk;
    const int64_t q = fresh ? (int64_t)b * Q + order[(int64_t)b * Q + l] : 0;
    const int64_t si = (int64_t)b * cap + (l - topk);
    for (int d = threadIdx.x; d < D; d += blockDim.x) out.emb[so * D + d] = fresh ? dec[q * D + d] : in.emb[si * D + d];
    if (threadIdx.x < 2) out.vel[so * 2 + threadIdx.x] = fresh ? bbox[q * ld_bbox + ld_bbox - 2 + threadIdx.x] : in.vel[si * 2 + threadIdx.x];
    if (threadIdx.x == 2) out.ts[so] = (fresh ? 0.0 : in.ts[si]) - timestamp[b];                    // :362,376
    if (threadIdx.x < 16) src_pose[threadIdx.x] = fresh ? rec_pose[q * 16 + threadIdx.x] : in.pose[si * 16 + threadIdx.x];
    if (threadIdx.x >= 32 && threadIdx.x < 35) src_ref[threadIdx.x - 32] = fresh ? bbox[q * ld_bbox + threadIdx.x - 32] : in.ref[si * 3 + threadIdx.x - 32];
    __syncthreads();
    if (threadIdx.x < 16) {                                                                         // :377 ego_pose @ pose
        const int i = threadIdx.x >> 2, j = threadIdx.x & 3;
        out.pose[so * 16 + threadIdx.x] = dot4(E + i * 4, src_pose[j], src_pose[4 + j], src_pose[8 + j], src_pose[12 + j]);
    }
    if (threadIdx.x >= 32 && threadIdx.x < 35) {                                                    // :375
        const int i = threadIdx.x - 32;
        out.ref[so * 3 + i] = dot4(E + i * 4, src_ref[0], src_ref[1], src_ref[2], 1.0f);
    }
}

}  // namespace

extern "C" {

int toc3d_memory_pre_update(float* emb, float* ref, double* ts, float* pose, float* vel, const float* prev_exists, const double* timestamp,
                            const float* ego_pose_inv, const float* pseudo_reference_points, const float* pc_range, int64_t B, int64_t capacity,
                            int64_t memory_len, int64_t num_propagated, int64_t embed_dims, int fresh, toc3d_stream_t stream) {
    TOC3D_REQUIRE(emb && ref && ts && pose && vel && prev_exists && timestamp && ego_pose_inv && pc_range, "toc3d_memory_pre_update: null buffer");
    TOC3D_REQUIRE(B > 0 && memory_len > 0 && capacity >= memory_len && embed_dims > 0 && num_propagated >= 0 && num_propagated <= memory_len &&
                  (num_propagated == 0 || pseudo_reference_points), "toc3d_memory_pre_update: bad dims");
    TOC3D_REQUIRE(B <= 65535, "toc3d_memory_pre_update: batch too large");
    toc3d_launch(pre_update_kernel, dim3((unsigned)memory_len, (unsigned)B), dim3(64), 0, as_stream(stream), Bank{emb, ref, ts, pose, vel},
                       prev_exists, timestamp, ego_pose_inv, pseudo_reference_points, pc_range, (int)B, (int)capacity, (int)memory_len,
                       (int)num_propagated, (int)embed_dims, fresh);
    TOC3D_LAUNCH_CHECK("toc3d_memory_pre_update");
    return TOC3D_OK;
}

int toc3d_memory_scores(const float* cls_scores, int64_t rows, int64_t num_classes, float* score, toc3d_stream_t stream) {
    TOC3D_REQUIRE(cls_scores && score && rows >= 0 && num_classes > 0, "toc3d_memory_scores: bad arguments");
    if (rows == 0) return TOC3D_OK;
    toc3d_launch(score_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream), cls_scores, rows, (int)num_classes, score);
    TOC3D_LAUNCH_CHECK("toc3d_memory_scores");
    return TOC3D_OK;
}

int toc3d_memory_post_update(const float* emb_in, const float* ref_in, const double* ts_in, const float* pose_in, const float* vel_in,
                             float* emb_out, float* ref_out, double* ts_out, float* pose_out, float* vel_out, const int64_t* order,
                             const float* rec_ego_pose, const float* bbox_preds, int64_t ld_bbox, const float* outs_dec, const float* ego_pose,
                             const double* timestamp, int64_t B, int64_t Q, int64_t capacity, int64_t memory_len, int64_t topk, int64_t embed_dims,
                             toc3d_stream_t stream) {
    TOC3D_REQUIRE(emb_in && ref_in && ts_in && pose_in && vel_in && emb_out && ref_out && ts_out && pose_out && vel_out && order && rec_ego_pose &&
                  bbox_preds && outs_dec && ego_pose && timestamp, "toc3d_memory_post_update: null buffer");
    TOC3D_REQUIRE(emb_in != emb_out && ref_in != ref_out && ts_in != ts_out && pose_in != pose_out && vel_in != vel_out,
                  "toc3d_memory_post_update: in and out banks must be distinct (the update shifts every slot)");
    TOC3D_REQUIRE(B > 0 && B <= 65535 && topk > 0 && topk <= Q && memory_len > 0 && capacity == memory_len + topk && embed_dims > 0 && ld_bbox >= 5,
                  "toc3d_memory_post_update: bad dims (capacity must be memory_len + topk)");
    toc3d_launch(post_update_kernel, dim3((unsigned)capacity, (unsigned)B), dim3(64), 0, as_stream(stream),
                       Bank{const_cast<float*>(emb_in), const_cast<float*>(ref_in), const_cast<double*>(ts_in), const_cast<float*>(pose_in), const_cast<float*>(vel_in)},
                       Bank{emb_out, ref_out, ts_out, pose_out, vel_out}, order, rec_ego_pose, bbox_preds, (int)ld_bbox, outs_dec, ego_pose, timestamp,
                       (int)B, (int)Q, (int)capacity, (int)topk, (int)embed_dims);
    TOC3D_LAUNCH_CHECK("toc3d_memory_post_update");
    return TOC3D_OK;
}

}  // extern "C"
