/*
 * libtoc3d_gfx950 -- C ABI of the MI355X (gfx950) kernels for the ToC3D / EVA-02 ViT backbone hot path.
 *
 * The reference (DYZhang09/ToC3D) is 100 % Python on stock torch ops: it has no native operator
 * interface for this path, so there is nothing like an existing FFI to bind.  Each entry point below
 * therefore replaces a *reference Python function* (cited as file:line, paths relative to
 * projects/mmdet3d_plugin/models/) and is what a maintainer would call from that function's body --
 * see INTEGRATION.md for the ctypes stubs.
 *
 * Conventions (all functions):
 *   - return TOC3D_OK (0) or a negative error code; never throw, never exit; text via toc3d_last_error().
 *   - the caller owns every buffer (device pointers); no hidden allocation, no hidden synchronisation;
 *     all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the legacy default stream).
 *   - the library holds no mutable global state: packed weights and index maps are plain caller-owned
 *     device buffers, so calls are re-entrant across host threads / streams and capturable in a hipGraph.
 *   - `dtype` selects the arithmetic type of activations and packed weights: TOC3D_BF16 (bf16 operands on
 *     v_mfma_f32_16x16x32_bf16, f32 accumulate) or TOC3D_F32 (exact f32 on v_mfma_f32_16x16x4_f32, the
 *     strict-parity path).  The residual stream, LayerNorm statistics, softmax, scorer and merge weights
 *     are always f32.
 *   - "act" buffers are row-major [rows, ld] of `dtype`; K-like dims are padded to multiples of 64
 *     elements with zeros (e.g. the SwiGLU hidden 2730 -> 2752).
 *   - the dlopen() of this library creates no HIP context (safe before fork()).
 */
#ifndef TOC3D_H_
#define TOC3D_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TOC3D_ABI_VERSION 8   /* bumped whenever entry points are added or changed; toc3d_amd/lib.py checks it before binding symbols */

#define TOC3D_OK 0
#define TOC3D_ERR_ARG (-1)
#define TOC3D_ERR_UNSUPPORTED (-2)
#define TOC3D_ERR_LAUNCH (-3)

/* dtype */
#define TOC3D_DTYPE_F32 0
#define TOC3D_DTYPE_BF16 1
/* Linear layers only (toc3d_linear*, toc3d_conv3x3_nhwc): buffers and packed weights exactly as for TOC3D_DTYPE_F32, but every product a.w is
 * formed on the bf16 matrix cores as hi.hi + hi.lo + lo.hi of the operands' (hi, lo) bf16 splits, f32 accumulate -- relative error of a product
 * <= ~2^-16 instead of exact, 3 bf16 MFMAs instead of 8 f32 ones per 16x16x32 step.  The "parity-grade fast" precision (precision="fp32x3" of the
 * host modules): every other kernel runs its f32 form. */
#define TOC3D_DTYPE_F32X3 2
/* ... and with a THREE-way split (hi, mid, lo = all 24 mantissa bits) and six products hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi: f32-grade
 * products (dropped terms <= 2^-26) from 6 bf16 MFMAs instead of 8 f32 ones (precision="fp32x6"). */
#define TOC3D_DTYPE_F32X6 3
/* TOC3D_DTYPE_F32X3 on (hi, lo) PLANES (round 4).  The F32X3 kernel splits every operand tile into its bf16 (hi, lo) pair in LDS: 9-15 % of a launch for the
 * weights, 26-36 % for both operands (profiles/r04_x3_split_cost.txt) -- work that depends on the operand, not on the launch.  A buffer "in planes" has the
 * size and leading dimension of the f32 buffer; element c of a row lives in the 128-byte group c / 32: hi = bf16(x) at byte 2 (c % 32), lo = bf16(x - hi) at
 * byte 64 + 2 (c % 32) (toc3d_x3_planes converts; the kernel's own split writes exactly this image into LDS, so results are bit-identical to F32X3).
 *   F32X3W: W in planes (packed once: toc3d_pack_* with TOC3D_DTYPE_F32, then toc3d_x3_planes in place); A and every output plain f32.
 *   F32X3P: W and A in planes; the outputs a later GEMM multiplies -- the SwiGLU epilogues' `out`, `out_act` -- are written as planes, all others (bias / GELU /
 *           residual outputs, statistics) stay f32.  toc3d_layernorm_rows / toc3d_rebase_layernorm_rows / toc3d_gather_merge_ln* take this dtype too: f32
 *           arithmetic, output rows written as planes (they produce the A operands).
 * toc3d_window_attention on f32 q|k|v: TOC3D_DTYPE_F32 = exact-f32 products, f32 output; F32X3WO = the same with the output rows as planes; F32X3 = both contractions
 * (q.k, p.v) as bf16 x 3 products -- f32 RoPE, softmax and accumulation; 48 instead of 256 matrix-core cycles per 16x16x32 step -- f32 output; F32X3P = those
 * products and the output as planes (what precision="fp32x3" launches). */
#define TOC3D_DTYPE_F32X3W 4
#define TOC3D_DTYPE_F32X3P 5
/* ... the two mixed forms: F32X3WO = W in planes, A plain f32, the GEMM-to-GEMM outputs as planes (a producer whose own A is not split yet);
 * F32X3WA = W and A in planes, every output plain f32. */
#define TOC3D_DTYPE_F32X3WO 6
#define TOC3D_DTYPE_F32X3WA 7

/* toc3d_linear epilogues */
#define TOC3D_EPI_BIAS 0      /* out(act)  = A.W^T + bias                                            */
#define TOC3D_EPI_RESIDUAL 1  /* out(f32)  = residual + (A.W^T + bias)        (+ optional raw capture) */
#define TOC3D_EPI_SWIGLU 2    /* out(act)  = silu(A.W1^T + b1) * (A.W2^T + b2), W packed interleaved   */
#define TOC3D_EPI_GELU 3      /* out(act)  = gelu_erf(A.W^T + bias)                                    */
#define TOC3D_EPI_SWIGLU_STATS 4  /* SWIGLU + per-row (sum, sum of squares) of the written hidden units  (toc3d_linear_fused) */
#define TOC3D_EPI_RESIDUAL_LN 5   /* RESIDUAL with a LayerNorm of the A rows folded into the epilogue    (toc3d_linear_fused) */
#define TOC3D_EPI_RESIDUAL_STATS 6    /* RESIDUAL + act-dtype copy of the output rows + their per-row statistics (toc3d_linear_fused) */
#define TOC3D_EPI_SWIGLU_STATS_LN 7   /* SWIGLU_STATS with a LayerNorm of the A rows folded into the epilogue   (toc3d_linear_fused) */
#define TOC3D_EPI_CONV3X3 8           /* out(f32) = conv3x3(NHWC act tensor) + bias as an implicit GEMM          (toc3d_conv3x3_nhwc) */
#define TOC3D_EPI_QKV_ROPE 9          /* out(act) = [rope(q) * scale | rope(k) | v] of the fused q|k|v projection  (toc3d_linear_qkv_rope) */

typedef void* toc3d_stream_t;

int toc3d_abi_version(void);
const char* toc3d_last_error(void); /* thread-local, valid until the next failing call on this thread */

/* ---------------------------------------------------------------------------------------------------
 * Linear layers (MFMA GEMM).  Replaces every F.linear / nn.Linear on the path:
 *   backbones/eva_vit.py:45-49 (SwiGLU w1,w2,w3), :97-99,115 (q/k/v/proj), toc3d_eva_vit.py:495-497,514,
 *   backbones/toc3d_utils.py:99-112 (scorer in_conv/out_conv), backbones/eva_utils.py:279-287 (patch conv
 *   as im2col GEMM), necks/cp_fpn.py:114-135 (1x1 lateral conv).
 * A [M, lda] act, W [ceil(N/128)*128, ldw] act (toc3d_pack_weight / toc3d_pack_swiglu), bias [N] f32 or NULL.
 * K must be a multiple of 64 (pad at pack time).  Epilogue-specific arguments:
 *   RESIDUAL: out f32 [M, ldo]; residual f32 [*, ldr] or NULL; residual row = residual_row_mod > 0 ?
 *             m % residual_row_mod : m (patch-embed adds abs-pos[m % T]); if rep_index (int32 [M]) is given, rows with
 *             rep_index[m] >= 0 (representative tokens, toc3d_eva_vit.py:452-453) also store the raw branch output
 *             (A.W^T + bias) to rep_out[rep_index[m], N] f32.
 *   SWIGLU:   N = 2*Hp packed columns, out act [M, ldo >= Hp]; hidden units >= n_valid are written as 0.
 */
int toc3d_linear(int dtype, int epilogue, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                 void* out, int64_t ldo, const float* residual, int64_t ldr, int64_t residual_row_mod,
                 float* rep_out, const int32_t* rep_index, int64_t M, int64_t N, int64_t K, int64_t n_valid,
                 toc3d_stream_t stream);

/* Same with an explicit tile / pipeline variant (0 = the heuristic toc3d_linear uses): 1 = 128x128 tile, 2-deep
 * LDS ring; 2 = 128x128, 3-deep; 3 = 128x128, 4-deep; 4 = 128x64, 3-deep; 5 = 128x64, 4-deep; 6 = 64x128, 3-deep;
 * 7 = 64x64, 4-deep; 8 = 128x128 single LDS buffer; 9 = 128x64, 2-deep; 10 = 64x128, 2-deep; 11/12 = 128x128 with 32-wide
 * K tiles, 2-/3-deep (bf16 only); 13 = 128x64 single buffer; 14 = 64x64, 2-deep; 15 = variant 8 limited to 128 registers;
 * 16/17 = 128x128 on 8 wavefronts, single / double buffer; 18/19 = 256x128 on 8 wavefronts, double / single buffer;
 * 20 = 128x256, 21 = 256x256 (8 wavefronts, double buffer); 22-27 = K tiles of 128 / 256 elements (22: 128x128 K128,
 * 23: 128x128 K256, 24: 64x128 K256, 25: 128x128 K128 double buffer, 26: 64x128 K128, 27: 64x64 K256).  (1-7, 9-14, 27:
 * 4 wavefronts); 28/29 = 128x128 on 8 wavefronts with a 3-/4-deep ring, 30/31 = the same with 32-wide K tiles 4-/6-deep
 * (bf16), 32 = 256x128 3-deep, 33 = 128x64 on 8 wavefronts 4-deep; 34-37 = 16-wavefront 256x128 / 256x256 / 128x256 tiles;
 * 38/39 = 128x128 on 8 wavefronts with 32-wide K tiles 2-/3-deep, 40-42 = 128x256 / 256x128 single buffer and K32 rings;
 * 43-48 = 128x192 and 128x96 tiles (43/44 single buffer, 45/46 double buffer, 47 = 128x192 with 32x96 per wavefront so that
 * it serves SWIGLU, 48 = 3-deep); 49/50 = 192x128 double / single buffer; 51 = variant 16 compiled for 64 registers (four workgroups per CU); 52/53 = 192x192 single / double buffer;
 * 60-63 = phased 256x256 / 256x128 / 128x256 / 128x128 tiles (bf16, one or two workgroups per CU, four phases per K-tile); 64-66 = register-pipelined 128x128 rings (bf16; 4- and 3-deep with
 * 8 wavefronts, 4-deep with 4: the next K step's fragments are read while the current one multiplies -- for launches that run one workgroup per CU).  A variant whose per-wavefront column slab is not a
 * multiple of 32 cannot serve EPI_SWIGLU (error TOC3D_ERR_UNSUPPORTED).  variant + 100 = the same tile with the per-XCD band
 * order (each XCD keeps its A row band in L2 and walks the W panels once); variant + 200 / + 300 = a 2-D partition of the tiles over the XCDs (4 row bands x 2 column
 * halves / 2 row bands x 4 column quarters: an XCD streams half / a quarter of W instead of all of it -- the wide-N and long-K GEMMs are bound by the L2-miss traffic
 * through the fabric, profiles/r03_xcd_order_sweep.txt).  Every variant accumulates K in the same order: outputs are bit-identical across variants. */
int toc3d_linear_ex(int dtype, int epilogue, int variant, const void* A, int64_t lda, const void* W, int64_t ldw,
                    const float* bias, void* out, int64_t ldo, const float* residual, int64_t ldr, int64_t residual_row_mod,
                    float* rep_out, const int32_t* rep_index, int64_t M, int64_t N, int64_t K, int64_t n_valid,
                    toc3d_stream_t stream);


/* SwiGLU.ffn_ln folded across the w1|w2 -> w3 boundary (backbones/eva_vit.py:47-49: x = ffn_ln(act(w1 x) * w2 x); x = w3 x), bf16 only.
 *   w3(LN(h))[n] = rstd * (h . W'[n] - mean * c1[n]) + c2[n],   W' = gamma (.) w3 (packed in the act dtype), c1[n] = sum_j W'[n, j],
 *   c2[n] = sum_j beta[j] w3[n, j] + b3[n]   (toc3d_pack_weight_lnfold), mean / rstd over the n_valid hidden units of the row.
 * So the LayerNorm pass over h (read + write of M x Hp activations per block, one launch) disappears:
 *   EPI_SWIGLU_STATS  = EPI_SWIGLU that also leaves per-row partial (sum h, sum h^2) of the *rounded* hidden units in stats_out:
 *                       int32 header [4] (header[0] = slots per row actually written) followed by f32 [M, stats_cap, 2]; one slot per
 *                       128 packed columns, every slot summed in one fixed tree whatever the tile variant (variants whose N-tile is not a
 *                       multiple of 128 cannot serve it: TOC3D_ERR_UNSUPPORTED), so results do not depend on the variant;
 *                       stats_out_cap >= ceil(N / 128).
 *   EPI_RESIDUAL_LN   = EPI_RESIDUAL on A = h with W = W', bias = c2, col_sums = c1, ln_n = number of valid hidden units, ln_eps;
 *                       reads stats_in (fixed summation order: independent of its own tile variant too).
 * The same identity removes norm2 (eva_vit.py:263, toc3d_eva_vit.py:381): the attention projection's residual epilogue leaves the updated
 * residual-stream rows in the act dtype plus their statistics, and the w1|w2 GEMM applies the LayerNorm to its accumulators:
 *   EPI_RESIDUAL_STATS   = EPI_RESIDUAL + out_act [M, ld_act] (act-dtype copy of the f32 output rows) + stats_out with one slot per 64 output
 *                          columns (N-tiles must be multiples of 64); stats_out_cap >= ceil(N / 64).
 *   EPI_SWIGLU_STATS_LN  = EPI_SWIGLU_STATS on A = out_act with W = gamma-scaled interleaved weights, bias = c2, col_sums = c1 in packed column
 *                          order (toc3d_pack_swiglu_lnfold); reads stats_in, writes stats_out (different buffers).
 * stats_out / stats_in: int32 header [4] + f32 [M, cap, 2] each; stats_in_cap may carry the number of slots per row the producer wrote in its
 * upper 32 bits (cap | slots << 32) when the host knows it, which saves the consumer the dependent read of the header; ln_n = width of the normalised rows (valid hidden units for EPI_RESIDUAL_LN,
 * K for EPI_SWIGLU_STATS_LN).  residual_index (int32 [M] or NULL, residual epilogues): output row m takes its residual from row
 * residual_index[m] of `residual` (a compact row whose f32 residual still sits in the token-major stream, toc3d_gather_merge_ln_ex with
 * kept_copy = 0), or, where residual_index[m] < 0, from output row m itself, read in place (representative rows).
 * EPI_CONV3X3 is reached through toc3d_conv3x3_nhwc below, which is this call with A = the NHWC tensor, lda = C, K = 9 * C, out_act = the zero
 * line and ld_act = h << 32 | w (hosts that tune the tile variant per shape call it in this form directly).
 * Every other argument as toc3d_linear_ex; epilogues 0-3 ignore the extra arguments. */
int toc3d_linear_fused(int dtype, int epilogue, int variant, const void* A, int64_t lda, const void* W, int64_t ldw,
                       const float* bias, void* out, int64_t ldo, const float* residual, int64_t ldr, int64_t residual_row_mod,
                       float* rep_out, const int32_t* rep_index, int64_t M, int64_t N, int64_t K, int64_t n_valid,
                       float* stats_out, int64_t stats_out_cap, const float* stats_in, int64_t stats_in_cap,
                       const float* col_sums, int64_t ln_n, float ln_eps, void* out_act, int64_t ld_act, const int32_t* residual_index,
                       toc3d_stream_t stream);

/* Deterministic split-K for the residual epilogues (TOC3D_EPI_RESIDUAL, _RESIDUAL_LN, _RESIDUAL_STATS: attn.proj eva_vit.py:115,262 / toc3d_eva_vit.py:514,379 and
 * mlp.w3 eva_vit.py:49,263 / toc3d_eva_vit.py:384 -- N = C = 1024 outputs per row, i.e. 136-376 tiles of 128x128 for a 6-view frame on a 256-CU chip).
 *   variant = 1000 * split + tile variant: `split` (2 .. TOC3D_SPLITK_MAX) workgroups per output tile, each multiplying its own range of K (cuts at multiples of
 *   128 elements, the same for every tile variant); tile variants with a split-K form: 1, 9, 10, 14, 16, 17, 19, 22, 26, 28, 29, 55, 56 (bf16; f32 / bf16 x 3: not 29, 55, 56).
 *   The partial accumulators meet in `workspace`; the workgroup that arrives last at a tile's ticket adds them IN SLICE ORDER (its own from registers) and runs the
 *   epilogue -- no atomics on data, no workgroup waits for another: outputs are bit-reproducible from run to run and identical for every tile variant of one
 *   `split` (they differ from the unsplit variants in the last bits: another order of the K sum).
 *   workspace: >= toc3d_linear_splitk_workspace_bytes(variant, M, N) bytes, 256-byte aligned; its first TOC3D_SPLITK_TICKET_BYTES hold the arrival tickets and must
 *   be ZERO before the first launch (every launch leaves them zero again); the rest is scratch.  Launches on one stream may share a workspace, launches that can
 *   run concurrently must not.  variant < 1000 = toc3d_linear_fused (workspace ignored). */
#define TOC3D_SPLITK_MAX 4
#define TOC3D_SPLITK_TICKET_BYTES 65536
int64_t toc3d_linear_splitk_workspace_bytes(int variant, int64_t M, int64_t N);      /* < 0: not a split-K variant */
int toc3d_linear_fused_ws(int dtype, int epilogue, int variant, const void* A, int64_t lda, const void* W, int64_t ldw,
                          const float* bias, void* out, int64_t ldo, const float* residual, int64_t ldr, int64_t residual_row_mod,
                          float* rep_out, const int32_t* rep_index, int64_t M, int64_t N, int64_t K, int64_t n_valid,
                          float* stats_out, int64_t stats_out_cap, const float* stats_in, int64_t stats_in_cap,
                          const float* col_sums, int64_t ln_n, float ln_eps, void* out_act, int64_t ld_act, const int32_t* residual_index,
                          void* workspace, int64_t workspace_bytes, toc3d_stream_t stream);

/* mlp.w1 / mlp.w2 interleaved as toc3d_pack_swiglu, scaled by norm2's gamma per input channel; c1 [2*Hp] = row sums of the ROUNDED scaled
 * weights, c2 [2*Hp] = beta . w + b, both in packed row order. */
int toc3d_pack_swiglu_lnfold(int dtype, const float* w1, const float* w2, const float* b1, const float* b2, const float* gamma, const float* beta,
                             int64_t Hd, int64_t K, void* out_w, float* c1, float* c2, int64_t Hp, int64_t Kp, toc3d_stream_t stream);
/* w3 f32 [N, K], ffn_ln gamma / beta f32 [K], b3 f32 [N] -> W' act [Np, Kp] (zero padded), c1 f32 [N] (sums of the ROUNDED W' rows, so
 * that the mean term cancels exactly what the GEMM accumulated), c2 f32 [N]. */
int toc3d_pack_weight_lnfold(int dtype, const float* w3, const float* gamma, const float* beta, const float* b3, int64_t N, int64_t K,
                             void* out_w, int64_t Np, int64_t Kp, float* c1, float* c2, toc3d_stream_t stream);

/* f32 rows [rows, K] (leading dim ld_src) -> (hi, lo) planes (TOC3D_DTYPE_F32X3W / F32X3P above) in dst (leading dim ld_dst, a multiple of 32; K a multiple of 32).
 * dst == src converts in place (equal leading dims). */
int toc3d_x3_planes(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int64_t K, toc3d_stream_t stream);

/* f32 [N, K] state-dict weight -> act [Np, Kp], zero padded (Np multiple of 128, Kp multiple of 64). */
int toc3d_pack_weight(int dtype, const float* w, int64_t N, int64_t K, void* out, int64_t Np, int64_t Kp, toc3d_stream_t stream);
/* mlp.w1 / mlp.w2 (eva_vit.py:35-36) -> interleaved [2*Hp, Kp] + bias [2*Hp]; packed row 32b+i = w1 row 16b+i,
 * row 32b+16+i = w2 row 16b+i. */
int toc3d_pack_swiglu(int dtype, const float* w1, const float* w2, const float* b1, const float* b2, int64_t Hd, int64_t K,
                      void* out_w, float* out_b, int64_t Hp, int64_t Kp, toc3d_stream_t stream);

/* PatchEmbed input side (backbones/eva_utils.py:283-287): img f32 NCHW [V, Cin, H, W] -> rows
 * [V*(H/p)*(W/p), ldo] act with column order (ch, py, px) == flattened Conv2d weight. */
int toc3d_im2col_patches(int dtype, const float* img, void* out, int64_t ldo, int64_t V, int64_t Cin, int64_t H, int64_t W,
                         int64_t patch, toc3d_stream_t stream);

/* Camera images as uint8.  img uint8 HWC [V, H, W, 3] (BGR as loaded,
 * mmdet3d/datasets/pipelines/loading.py:47-50); mean3 / std3 are HOST float[3] (img_norm_cfg, configs/ToC3D/ToC3D_faster.py:13-14).
 * toc3d_normalize_images = NormalizeMultiviewImage (datasets/pipelines/transform_3d.py:87-100, mmcv.imnormalize:
 *   channel flip if to_rgb, then fl32(fl32(x - mean) * fl32(1/std))) + PadMultiViewImage (transform_3d.py:38-50: zeros bottom /
 *   right up to [Hp, Wp]) + the HWC->CHW transpose of DefaultFormatBundle (mmdet3d/datasets/pipelines/formating.py:42-47)
 *   -> out f32 NCHW [V, 3, Hp, Wp], i.e. the tensor ToC3DEVAViT.forward receives.
 * toc3d_im2col_patches_u8 = the same fused into toc3d_im2col_patches: rows [V*(Hp/p)*(Wp/p), ldo], identical values. */
int toc3d_normalize_images(const uint8_t* img, int64_t V, int64_t H, int64_t W, const float* mean3, const float* std3, int to_rgb,
                           float* out, int64_t Hp, int64_t Wp, toc3d_stream_t stream);
int toc3d_im2col_patches_u8(int dtype, const uint8_t* img, int64_t V, int64_t H, int64_t W, const float* mean3, const float* std3,
                            int to_rgb, void* out, int64_t ldo, int64_t Hp, int64_t Wp, int64_t patch, toc3d_stream_t stream);

/* get_abs_pos (backbones/eva_utils.py:229-258): pos f32 [S*S, C] (cls row already dropped) -> out f32 [h*w, C],
 * bicubic, align_corners=False, A = -0.75 (torch F.interpolate semantics).  Copy if S == h == w. */
int toc3d_abs_pos_bicubic(const float* pos, int64_t S, int64_t C, float* out, int64_t h, int64_t w, toc3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * LayerNorm.
 * toc3d_layernorm_rows: Block.norm1 / norm2 (backbones/eva_vit.py:249,263; toc3d_eva_vit.py:372,382) and the
 *   scorer's in_conv LayerNorm (toc3d_utils.py:100,118-122).  x f32 [*, ldx]; row m reads x[row_index ?
 *   row_index[m] : m] (index < 0 = all-zero row -> output beta); row_scale (nullable) multiplies the row
 *   first (token mask, toc3d_utils.py:118).  out act [M, ldo].  One wavefront per row, two-pass variance.
 * toc3d_layernorm_act: SwiGLU.ffn_ln over the hidden (eva_vit.py:39,48).  x act [M, ldx], n valid columns;
 *   out act [M, ldo]; columns [n, ldo) are written as zeros (K padding for w3).
 */
int toc3d_layernorm_rows(int dtype, const float* x, int64_t ldx, const int32_t* row_index, const float* row_scale,
                         const float* gamma, const float* beta, float eps, void* out, int64_t ldo, int64_t M, int64_t C,
                         toc3d_stream_t stream);
int toc3d_layernorm_act(int dtype, const void* x, int64_t ldx, const float* gamma, const float* beta, float eps, void* out,
                        int64_t ldo, int64_t M, int64_t n, toc3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Window geometry for dense blocks: window_partition (backbones/eva_utils.py:89-110) as index maps, no copy.
 * For V views of h x w tokens and window side L: nW = V*ceil(h/L)*ceil(w/L) windows of N = L*L slots.
 *   rows  [nW, N] int32: global token row (v*h*w + r*w + c) of the j-th *real* token of the window (slot order)
 *   slots [nW, N] int32: its slot id sr*L + sc (row of the RoPE table, eva_utils.py:364-371)
 *   count [nW]: number of real tokens;  npad [nW]: N - count zero-padded slots (keys k=0, v=v_bias)
 */
int toc3d_window_map_dense(int64_t V, int64_t h, int64_t w, int64_t L, int32_t* rows, int32_t* slots, int32_t* count,
                           int32_t* npad, toc3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Windowed multi-head attention with RoPE-by-slot (backbones/eva_vit.py:101-113, toc3d_eva_vit.py:499-512,
 * backbones/eva_utils.py:378-379,396-403).  head_dim must be 64.
 * qkv act [*, ldqkv] = [q | k | v] per row (each C wide, head-major); for window i the participating rows are
 * rows[i*stride + j], j < count[i], with RoPE table rows slots[i*stride + j]; q,k are rotated (pairs 2t,2t+1),
 * q scaled by `scale` after RoPE, softmax over the window's keys plus npad[i] virtual keys with logit 0 and
 * value v_bias (NULL npad = none; dense blocks, where zero-padding follows LayerNorm).  Accelerated blocks pad *before*
 * LayerNorm (toc3d_eva_vit.py:414,372), so a kept padded slot is the constant row LN(0)=beta: with count_k (keys per
 * window, >= count) the entries j >= count[i] may carry rows = -1 and take q|k|v from pad_qkv (act [3C], the q|k|v
 * projection of beta computed at pack time) with RoPE slot slots[i*stride+j]; they are keys only (their outputs are
 * dropped by window_unpartition).  out act [*, ldo]: out[row, head*64 + d].  Flash-style, keys streamed in tiles of 64
 * through LDS, online softmax in f32.  max_count = max_i count[i] (grid sizing only).
 * rope_cos/rope_sin are the reference's buffers freqs_cos/freqs_sin [rope_side^2, 64]; the kernel relies on their axial
 * structure (VisionRotaryEmbeddingFast, eva_utils.py:362-371: dims 0..31 depend on slot / rope_side, dims 32..63 on
 * slot % rope_side, each frequency repeated for the pair (2i, 2i+1)) and keeps a [2, rope_side, 16] extract in LDS;
 * the host module verifies that structure when it packs the buffers.
 */
int toc3d_window_attention(int dtype, const void* qkv, int64_t ldqkv, void* out, int64_t ldo, const int32_t* rows,
                           const int32_t* slots, const int32_t* count, const int32_t* count_k, const int32_t* npad,
                           const void* pad_qkv, int64_t stride, int64_t nwin, int64_t max_count, int64_t num_heads,
                           const float* rope_cos, const float* rope_sin, int64_t rope_side, const float* v_bias, float scale,
                           toc3d_stream_t stream);
/* The same launch with a weight prefetch riding on it: up to 4 read-only device buffers (HOST arrays prefetch_ptrs / prefetch_bytes of
 * n_prefetch entries, 16-byte aligned) are streamed through the caches by `prefetch_workgroups` extra workgroups (0 = 64) of the attention
 * grid and discarded -- the packed weights of the GEMMs that follow (this block's proj / w1|w2 / w3, the next block's q|k|v).  The attention
 * kernels are latency-bound and leave HBM idle; the weights were last touched a frame ago.  Results are identical to toc3d_window_attention. */
int toc3d_window_attention_pf(int dtype, const void* qkv, int64_t ldqkv, void* out, int64_t ldo, const int32_t* rows,
                              const int32_t* slots, const int32_t* count, const int32_t* count_k, const int32_t* npad,
                              const void* pad_qkv, int64_t stride, int64_t nwin, int64_t max_count, int64_t num_heads,
                              const float* rope_cos, const float* rope_sin, int64_t rope_side, const float* v_bias, float scale,
                              int64_t n_prefetch, const void* const* prefetch_ptrs, const int64_t* prefetch_bytes, int64_t prefetch_workgroups,
                              toc3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Token selection.
 * toc3d_rank_desc: ScoreBasedTokenSelector.sample's sort (backbones/toc3d_utils.py:139) with the tie rule pinned
 *   (descending, equal scores -> lower index first == torch.sort(stable=True)).  scores f32 [B, n] ->
 *   order int64 [B, n] (order[b, rank] = index).  keep_idx = order[:, :k], drop_idx = order[:, k:].
 * toc3d_window_topk: the per-window selection of an accelerated block (backbones/toc3d_eva_vit.py:412-438):
 *   partitions the image-level scores f32 [V, h, w] into windows of side L (pad score -1e6, :415), ranks the
 *   N = L*L slots of every window, keeps k = int(N*ratio) < N (computed by the caller) and emits
 *     order  [nW, N] int32   slot ids, descending-stable; [:k] slow, [k:] fast
 *     tok    [nW, N] int32   global token row of order[.,j], or -1 for a padded slot
 *     wgt    [nW, N] f32     merge_tokens weight s_j / sum_{fast} s (toc3d_utils.py:68) for j >= k, 0 for j < k
 *   and the *compact kept set*.  A kept padded slot is the same row for every window (x = 0 -> LN(0) = beta) and its
 *   block output is discarded, so only real kept tokens get rows: window i owns cap_i = min(k, real_i) + 1 rows
 *   [off_i, off_i + cap_i) = kept real tokens in sorted order (, explicit zero rows in the pathological case that a
 *   real token scored <= -1e6), representative token last; total rows = toc3d_window_topk_rows(V,h,w,L,k).
 *     prow      [nW, N] int32   compact row of sorted position p < k (-1: virtual pad)
 *     crow_tok  [rows]  int32   source token of each compact row (-1 zero row, -2 representative token)
 *     rep_index [rows]  int32   window id for representative rows, else -1 (toc3d_linear rep capture)
 *     rep_row   [nW]    int32   compact row of the window's representative token
 *     arows/aslots [nW, k+1] int32  attention key list: compact rows first (queries = the first acount_q[i] = cap_i
 *                      entries; representative token uses RoPE slot k, :434), then the kept pads as virtual keys
 *                      (arows = -1, aslots = their slot);  acount_k[i] = k + 1
 *     crow_rc   [rows]  int32   (optional, NULL = not wanted) RoPE position of each compact row as (slot / L) << 16 | (slot % L),
 *                      representative rows at slot k: the per-row table toc3d_linear_qkv_rope consumes
 */
int toc3d_rank_desc(const float* scores, int64_t B, int64_t n, int64_t* order, toc3d_stream_t stream);
int64_t toc3d_window_topk_rows(int64_t V, int64_t h, int64_t w, int64_t L, int64_t k);
int toc3d_window_topk(const float* scores, int64_t V, int64_t h, int64_t w, int64_t L, int64_t k, int32_t* order,
                      int32_t* tok, float* wgt, int32_t* prow, int32_t* crow_tok, int32_t* rep_index, int32_t* rep_row,
                      int32_t* arows, int32_t* aslots, int32_t* acount_q, int32_t* acount_k, int32_t* crow_rc, toc3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Pre-rotated attention (bf16 path, round 3; bf16 x 3 on (hi, lo) planes, round 6): RoPE and the q scale move from the attention kernel into the q|k|v projection's
 * epilogue, so the attention kernel stages K and V by DMA (global_load_lds) with no arithmetic at all.  dtype = TOC3D_DTYPE_BF16, or for precision "fp32x3":
 * toc3d_linear_qkv_rope with TOC3D_DTYPE_F32X3P (A and W in planes) / F32X3WO (A plain f32, W in planes) writes q | k | v as planes (ldo % 32 == 0, 128-byte aligned), and
 * toc3d_window_attention_rot with TOC3D_DTYPE_F32X3P reads them and writes `out` as planes (leading dimensions in f32 elements; both contractions as bf16 x 3 products,
 * f32 softmax statistics and accumulation; windows of up to 1024 keys -- over 288 keys in 128-key super-tiles; no weight prefetch on this form).
 * toc3d_linear_qkv_rope: out act [M, ldo >= N] = [rope(q) * q_scale | rope(k) | v] for the fused projection W = [q; k; v] packed by
 *   toc3d_pack_weight (N = 3C, heads of 64 dims; backbones/eva_vit.py:97-109, toc3d_eva_vit.py:495-508, eva_utils.py:378-379).  Row m is rotated by
 *   the table rows rope_rc[m] = r << 16 | c: dims 0..31 of every head by row r of the first table half, dims 32..63 by row c of the second
 *   (VisionRotaryEmbeddingFast's axial layout, eva_utils.py:362-371; r = slot / rope_side, c = slot % rope_side for window slot `slot`);
 *   rope_tab holds the compact tables f32 [cos | sin][2, rope_side <= 64, 16] (one entry per frequency pair; extracted from the module's
 *   freqs_cos / freqs_sin buffers by the host; 16-byte aligned: the kernel copies it into LDS by DMA while its K loop runs).  The rotation runs on the f32 accumulators (+ bias): one rounding to bf16 instead of two.
 * toc3d_window_attention_rot: toc3d_window_attention_pf on such a buffer.  Same window lists (rows / count / count_k / npad, stride <= 416);
 *   virtual kept-pad keys (rows[j] < 0) read row slots[j] of pad_rot [window slots, ldqkv] -- the projection of LN(0) = beta rotated for every
 *   window slot, produced at pack time by toc3d_linear_qkv_rope itself (identical bits to an explicit pad row).  One workgroup per (window, head)
 *   holds the window's K and V of that head in LDS; scores transposed, P in registers, V read with ds_read_b64_tr_b16.  The softmax is EXP2-BASED (round 5):
 *   pass q_scale = head_dim^-0.5 * log2(e) to toc3d_linear_qkv_rope for buffers this kernel reads (softmax(s) = exp2(s log2 e - m) / sum); one pass over the keys,
 *   online softmax whose reference point moves only when a 32-key chunk's maximum exceeds it by more than 8 (deferred rescale): exact softmax up to rounding.  The prefetch buffers
 *   (as for toc3d_window_attention_pf; prefetch_workgroups != 0 enables them) are pulled through the caches by the attention wavefronts
 *   themselves, a few KB each by LDS-DMA while they compute -- no extra workgroups.
 */
float toc3d_attn_rot_q_scale(int64_t head_dim);   /* the q_scale to pass for buffers toc3d_window_attention_rot reads: head_dim^-0.5 * log2(e) */
int toc3d_linear_qkv_rope(int dtype, int variant, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo,
                          int64_t M, int64_t N, int64_t K, const int32_t* rope_rc, const float* rope_tab, int64_t rope_side,
                          float q_scale, toc3d_stream_t stream);
int toc3d_window_attention_rot(int dtype, const void* qkv, int64_t ldqkv, void* out, int64_t ldo, const int32_t* rows, const int32_t* slots,
                               const int32_t* count, const int32_t* count_k, const int32_t* npad, const void* pad_rot, int64_t stride,
                               int64_t nwin, int64_t max_count, int64_t num_heads, const float* v_bias,
                               int64_t n_prefetch, const void* const* prefetch_ptrs, const int64_t* prefetch_bytes, int64_t prefetch_workgroups,
                               toc3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Accelerated block front / back end (backbones/toc3d_eva_vit.py:421-430 and :449-467).
 * toc3d_gather_merge_ln: builds the compact kept set from the residual stream x f32 [V*T, C]:
 *   ordinary rows: copy of token crow_tok[r] (zeros for -1) -> shortcut f32 [rows, C] and LN1 -> a_out act;
 *   representative rows rep_row[i]: sum_{j>=k} wgt[i,j] * x[tok[i,j]] (batch_index_select + merge_tokens,
 *              toc3d_utils.py:28-44,65-70), same two outputs.  One launch; merge reduced deterministically.
 * toc3d_scatter_update: batch_index_fill + window_unpartition (toc3d_utils.py:47-62, eva_utils.py:113-133) done
 *   in place on x: kept tokens <- slow_out[prow]; dropped tokens += rep_raw1[i] + rep_raw2[i] (:452-456); padded
 *   slots are dropped.  rep_raw3 / rep_raw4 (nullable pair): the two updates of a second block that ran on the same kept set
 *   without scattering in between (consecutive blocks of one window type select identical tokens), added after the first two.
 */
int toc3d_gather_merge_ln(int dtype, const float* x, int64_t C, const int32_t* tok, const float* wgt, const int32_t* crow_tok,
                          const int32_t* rep_row, int64_t nW, int64_t N, int64_t k, int64_t rows, const float* gamma,
                          const float* beta, float eps, float* shortcut, void* a_out, int64_t lda, toc3d_stream_t stream);
/* The same with kept_copy = 0: the f32 `shortcut` copy of the kept rows is NOT written (40 % of the kernel's bytes); only the representative
 * rows are.  The projection GEMM then reads the residual of a kept row from x through crow_tok (toc3d_linear_fused residual_index). */
int toc3d_gather_merge_ln_ex(int dtype, const float* x, int64_t C, const int32_t* tok, const float* wgt, const int32_t* crow_tok,
                             const int32_t* rep_row, int64_t nW, int64_t N, int64_t k, int64_t rows, const float* gamma,
                             const float* beta, float eps, float* shortcut, void* a_out, int64_t lda, int64_t kept_copy, toc3d_stream_t stream);
/* toc3d_gather_merge_ln_ex with every window's merge cut over `split` = 2, 4, 8 or 16 workgroups (0 = the default, 4) (the single-workgroup merge is bound by what one CU can load: 0.5-1 MB of
 *   dropped rows per window): same arguments, same results BIT FOR BIT (the partial sums are formed and added in the order of the single-workgroup
 *   kernel), plus `scratch` = toc3d_gather_merge_ln_scratch_bytes(nW, C) bytes of device memory, 256-byte aligned, zeroed ONCE by the caller (arrival
 *   counters + f32 partials; the kernel re-arms the counters itself, so a recorded launch plan replays it without a memset).  One scratch buffer per
 *   stream that may run the kernel; layout = [TOC3D_GATHER_SPLIT_COUNTER_BYTES of counters | partials], the same for every nW, so launches of different
 *   window counts on one stream may share a buffer sized for the largest. */
#define TOC3D_GATHER_SPLIT_COUNTER_BYTES 16384
int64_t toc3d_gather_merge_ln_scratch_bytes(int64_t nW, int64_t C);
int toc3d_gather_merge_ln_split(int dtype, const float* x, int64_t C, const int32_t* tok, const float* wgt, const int32_t* crow_tok,
                                const int32_t* rep_row, int64_t nW, int64_t N, int64_t k, int64_t rows, const float* gamma,
                                const float* beta, float eps, float* shortcut, void* a_out, int64_t lda, int64_t kept_copy,
                                void* scratch, int64_t scratch_bytes, int64_t split, toc3d_stream_t stream);
/* toc3d_rebase_layernorm_rows: norm1 (toc3d_eva_vit.py:372) of a block that continues on the previous block's compact rows instead
 *   of re-gathering them (shortcut rows f32 [rows, C], in place; LN -> out act [rows, ldo]).  Representative rows (rep_index[r] = window
 *   i >= 0) are first turned into what merge_tokens (toc3d_utils.py:65-70) would produce from the updated dropped tokens:
 *   slow[r] -= (1 - W_i) * (rep_raw1[i] + rep_raw2[i]), W_i = sum of wgt over the window's real dropped tokens. */
int toc3d_rebase_layernorm_rows(int dtype, float* slow, int64_t C, const int32_t* rep_index, const int32_t* tok, const float* wgt, int64_t N,
                                int64_t k, const float* rep_raw1, const float* rep_raw2, const float* gamma, const float* beta, float eps,
                                void* out, int64_t ldo, int64_t rows, toc3d_stream_t stream);
int toc3d_scatter_update(float* x, int64_t C, const int32_t* tok, const int32_t* prow, int64_t nW, int64_t N, int64_t k,
                         const float* slow_out, const float* rep_raw1, const float* rep_raw2, const float* rep_raw3, const float* rep_raw4,
                         toc3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Motion-aware query-guided scorer (backbones/toc3d_utils.py:232-252, 334-360; utils/misc.py:154-200;
 * utils/positional_encoding.py:14-81).
 * toc3d_motion_queries: get_motion_aware_queries for B frames x Q queries (query dim 256), for n_stages scorers in one
 *   launch (the three scorers see identical inputs but own separate weights: packed weight sets w + s*w_stride, outputs
 *   out + s*B*Q*256):
 *   queries f32 [B,Q,256], ref_points f32 [B,Q,3], vel f32 [B,Q,2], timestamp f64 [B,Q] (timestamp_is_f64=1) or
 *   f32, ego_pose f32 [B,Q,4,4], ego_pose_inv f32 [B,4,4]; `w` = the scorer's parameters packed by
 *   toc3d_motion_weights_floats()/toc3d_pack_motion_weights (transposed [in,out] for coalescing; dimt3 [128] and
 *   dimt1 [256] are the f32 tables temperature ** (2*floor(i/2)/n) of positional_encoding.py:18,32, computed by
 *   the host with the same torch expression because a 1-ulp change flips sin(2 pi t / dim_t) at epoch-scale t)
 *   -> out f32 [B,Q,256].
 *   sin/cos use the accurate range-reduced device functions (f64 for the time embedding), SURVEY quirk 14.
 * toc3d_collapse_query_scorer: input_proj + einsum + aggregate collapse exactly to a [C,2] matrix per frame:
 *   Wc = W_in^T (mq^T W_agg^T) * scale, bc = b_in (mq^T W_agg^T) * scale + b_agg.   wc f32 [B, C, 2], bc f32 [B, 2].
 * toc3d_score_tokens: per token: logits = (x*mask) . Wc + bc; pred = log_softmax(logits); score = pred[0];
 *   new soft mask = softmax(pred + gumbel)[0] (toc3d_utils.py:147, F.gumbel_softmax tau=1 with injected noise;
 *   NULL gumbel = zeros).  x f32 [V*T, C]; mask f32 [V*T] or NULL (= ones); views_per_frame maps view -> frame.
 *   Outputs pred f32 [V*T, 2], score f32 [V*T], mask_out f32 [V*T].
 */
int64_t toc3d_motion_weights_floats(void);
int toc3d_pack_motion_weights(const float* qe0_w, const float* qe0_b, const float* qe2_w, const float* qe2_b,
                              const float* pe_red_w, const float* pe_red_b, const float* pe_gam_w, const float* pe_gam_b,
                              const float* pe_bet_w, const float* pe_bet_b, const float* q_red_w, const float* q_red_b,
                              const float* q_gam_w, const float* q_gam_b, const float* q_bet_w, const float* q_bet_b,
                              const float* te_w, const float* te_b, const float* te_ln_w, const float* te_ln_b,
                              const float* pc_range, const float* dimt3, const float* dimt1, float* out,
                              toc3d_stream_t stream);
int toc3d_motion_queries(const float* w, int64_t n_stages, int64_t w_stride, const float* queries, const float* ref_points,
                         const float* vel, const void* timestamp, int timestamp_is_f64, const float* ego_pose,
                         const float* ego_pose_inv, int64_t B, int64_t Q, float* out, toc3d_stream_t stream);
int toc3d_collapse_query_scorer(const float* mq, const float* w_in, const float* b_in, const float* w_agg, const float* b_agg,
                                int64_t B, int64_t Q, int64_t C, float scale, float* wc, float* bc, toc3d_stream_t stream);
int toc3d_score_tokens(const float* x, int64_t C, const float* mask, const float* wc, const float* bc, const float* gumbel,
                       int64_t V, int64_t T, int64_t views_per_frame, float* pred, float* score, float* mask_out,
                       toc3d_stream_t stream);

/* Gumbel noise for the soft masks when the caller injects none (backbones/toc3d_utils.py:145-147: F.gumbel_softmax draws -log(E), E ~ Exp(1), in
 * eval too): out f32 [n] = -log(-log(U)), U from Philox4x32-10 keyed by `seed` with counter (state[0], element / 4).  state = uint64 [2] in device
 * memory, zero-initialised by the caller: state[0] is the frame counter, advanced by ONE per launch by the kernel itself (state[1] is its
 * ticket), so a recorded launch plan draws fresh, reproducible noise on every replay without any host-side argument changing. */
int toc3d_gumbel_noise(float* out, int64_t n, uint64_t seed, uint64_t* state, toc3d_stream_t stream);
/* The map toc3d_gumbel_noise applies to every 32-bit Philox word, on caller-supplied words: out[i] = -log(-log(U)), U = ((bits[i] >> 9) + 0.5) * 2^-23
 * in [2^-24, 1 - 2^-24] -- finite for EVERY word, 0 and 0xFFFFFFFF included (what the parity tests pin). */
int toc3d_gumbel_from_bits(const uint32_t* bits, int64_t n, float* out, toc3d_stream_t stream);

/* First-frame scorer pieces (ScoreBasedTokenSelector.score, backbones/toc3d_utils.py:114-129): the two big
 * Linear layers run through toc3d_linear (GELU epilogue); these cover the rest.
 * toc3d_global_mean_half: t act [V*T, ld]: columns [C/2, C) of every row of a view are replaced by that view's
 *   mean over tokens (:125-126).  toc3d_score_head: final Linear(256->2)+LogSoftmax (:110-111) fused with the
 *   Gumbel soft mask; f act [V*T, ld] (K = kdim), w f32 [2, kdim], b f32 [2]; outputs as toc3d_score_tokens. */
int toc3d_global_mean_half(int dtype, void* t, int64_t ld, int64_t V, int64_t T, int64_t C, toc3d_stream_t stream);
int toc3d_score_head(int dtype, const void* f, int64_t ld, int64_t kdim, const float* w, const float* b, const float* gumbel,
                     int64_t M, float* pred, float* score, float* mask_out, toc3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Layout helpers at the module boundary.
 * toc3d_nhwc_to_nchw: x f32 [V, T, C] -> out f32 [V, C, T] (materialises the reference's permute(0,3,1,2),
 *   toc3d_eva_vit.py:294, for consumers that need contiguous NCHW).
 * toc3d_im2col_3x3: CPFPN's 3x3 conv (necks/cp_fpn.py:124-133) as a GEMM: x f32 [V, h, w, C] (NHWC) ->
 *   rows [V*h*w, 9*C] act with column order (ky, kx, c); the weight is packed to match by the host.
 */
int toc3d_nhwc_to_nchw(const float* x, float* out, int64_t V, int64_t T, int64_t C, toc3d_stream_t stream);
/* CPFPN's 3x3 conv (pad 1, necks/cp_fpn.py:124-133) as an IMPLICIT GEMM: x act [V, h, w, C] (NHWC, C a multiple of 64), W act packed
 * [ceil128(Cout), 9*C] in (ky, kx, c) column order (the layout toc3d_im2col_3x3 produces rows for), bias f32 [Cout] or NULL ->
 * out f32 [V*h*w, ldo].  The GEMM's operand loader gathers the nine neighbours of every pixel itself (out-of-image taps read `zeros`, a
 * device buffer of >= 16 zero bytes), so the [V*h*w, 9*C] im2col matrix is never written or read; same K order as toc3d_im2col_3x3 +
 * toc3d_linear, hence the same bits.  `variant` as toc3d_linear_ex (phased variants excluded). */
int toc3d_conv3x3_nhwc(int dtype, int variant, const void* x, int64_t C, const void* W, int64_t ldw, const float* bias, float* out, int64_t ldo,
                       int64_t V, int64_t h, int64_t w, int64_t Cout, const void* zeros, toc3d_stream_t stream);
int toc3d_im2col_3x3(int dtype, const float* x, void* out, int64_t ldo, int64_t V, int64_t h, int64_t w, int64_t C,
                     toc3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Temporal memory bank: the head-side producer of the scorer inputs temp_queries / temp_ref_points /
 * temp_vel / temp_timestamp / temp_ego_pose (detectors/petr3d.py:115-134).  State = five caller-owned device buffers of
 * `capacity` = memory_len + topk slots per sample: emb f32 [B, cap, D], ref f32 [B, cap, 3], ts f64 [B, cap], pose f32
 * [B, cap, 4, 4], vel f32 [B, cap, 2].
 * toc3d_memory_pre_update: StreamPETRHead.pre_update_memory (dense_heads/streampetr_head.py:322-346) in place on slots
 *   [0, memory_len): += timestamp, ego_pose_inv @ pose, transform_reference_points, memory_refresh by prev_exists [B],
 *   then the pseudo reference points / identity poses of the first num_propagated slots.  fresh != 0: the bank is all zeros
 *   (reset_memory, :315-320, then :326-331) and only the last step applies.
 * toc3d_memory_scores: sigmoid(cls).topk(1).values (:361) -> score f32 [rows]; rank it with toc3d_rank_desc.
 * toc3d_memory_post_update: post_update_memory (:355-377): slots [0, topk) = the top-k queries (order from toc3d_rank_desc,
 *   lowest index first on ties) of the last decoder layer (bbox_preds [B, Q, ld_bbox]: ref = cols 0..2, velocity = last 2;
 *   outs_dec [B, Q, D]; rec_ego_pose [B, Q, 4, 4]), slots [topk, cap) = in-bank slots [0, memory_len); then ego_pose
 *   transform of the reference points and poses and -= timestamp.  In / out banks must be different buffers. */
int toc3d_memory_pre_update(float* emb, float* ref, double* ts, float* pose, float* vel, const float* prev_exists, const double* timestamp,
                            const float* ego_pose_inv, const float* pseudo_reference_points, const float* pc_range, int64_t B, int64_t capacity,
                            int64_t memory_len, int64_t num_propagated, int64_t embed_dims, int fresh, toc3d_stream_t stream);
int toc3d_memory_scores(const float* cls_scores, int64_t rows, int64_t num_classes, float* score, toc3d_stream_t stream);
int toc3d_memory_post_update(const float* emb_in, const float* ref_in, const double* ts_in, const float* pose_in, const float* vel_in,
                             float* emb_out, float* ref_out, double* ts_out, float* pose_out, float* vel_out, const int64_t* order,
                             const float* rec_ego_pose, const float* bbox_preds, int64_t ld_bbox, const float* outs_dec, const float* ego_pose,
                             const double* timestamp, int64_t B, int64_t Q, int64_t capacity, int64_t memory_len, int64_t topk, int64_t embed_dims,
                             toc3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Token side of StreamPETRHead.forward (second half; dense_heads/streampetr_head.py:378-422,627-639): the
 * consumers of the neck's features.  Linear layers run on toc3d_linear; these are the kernels around them.
 * toc3d_head_frustum_inputs: position_embeding up to the position_encoder's input.  For V = B*N views of h x w tokens and D depth
 *   bins (coords_d [D], device): pixel centres (misc.py:70-77) * depth through img2lidar [V, 4, 4] (= lidar2img^-1, device),
 *   normalised by position_range (HOST float[6]), inverse_sigmoid -> pos_in act [V*h*w, ld_pos] (3*D valid columns, the rest
 *   untouched: pre-zero the buffer once); cone f32 [V*h*w, 8] = [|fx|, |fy|] / 1e3 of camera (token % N) -- the reference's repeat
 *   order, :385-386 -- then the normalised points of the last depth bin and of bin D-30 (:419-420), also as act rows (ld_cone).
 * toc3d_relu_inplace: nn.ReLU between the Linear pairs.  toc3d_nchw_to_rows: neck output NCHW f32 -> [V*hw, ldo] act rows (:629).
 * toc3d_mln_apply: MLN.forward (models/utils/misc.py:181-188): gamma * LayerNorm(x, no affine, eps 1e-5) + beta -> f32 + act copy.
 * toc3d_se_gate: SELayer_Linear (misc.py:151): out = pos * sigmoid(se). */
int toc3d_head_frustum_inputs(int dtype, const float* img2lidar, const float* intrinsics, const float* coords_d, const float* position_range,
                              int64_t B, int64_t N, int64_t h, int64_t w, int64_t D, int64_t stride, int64_t pad_h, int64_t pad_w,
                              void* pos_in, int64_t ld_pos, void* cone_act, int64_t ld_cone, float* cone, toc3d_stream_t stream);
int toc3d_relu_inplace(int dtype, void* x, int64_t n, toc3d_stream_t stream);
int toc3d_nchw_to_rows(int dtype, const float* x, void* out, int64_t ldo, int64_t V, int64_t C, int64_t hw, toc3d_stream_t stream);
int toc3d_mln_apply(int dtype, const float* x, const float* gamma, const float* beta, int64_t M, int64_t E, float* out, void* out_act, int64_t ld_act,
                    toc3d_stream_t stream);
int toc3d_se_gate(const float* pos, const float* se, float* out, int64_t n, toc3d_stream_t stream);

/* Plain device-to-device copy as a kernel (recordable into a launch plan, unlike hipMemcpyAsync). */
int toc3d_copy_bytes(void* dst, const void* src, int64_t nbytes, toc3d_stream_t stream);
/* Up to 16 small device-to-device copies in ONE launch (every launch costs ~5 us of device time, and a frame has nine per-frame input
 * tensors to stage: temp_queries ... ego_pose_inv of detectors/petr3d.py:115-134 plus the three Gumbel tensors).  dst / src / nbytes are
 * HOST arrays of n entries (read during the call); device pointers need no alignment (16-byte body when both are aligned). */
int toc3d_copy_segments(int64_t n, void* const* dst, const void* const* src, const int64_t* nbytes, toc3d_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Launch plans: the per-frame host loop (the block loop of ToC3DEVAViT.forward, backbones/toc3d_eva_vit.py:263-291, and of
 * EVA_ViT.forward, backbones/eva_vit.py:419-426; CPFPN.forward, necks/cp_fpn.py:156-208) recorded once and replayed from C.
 * All shapes and buffers of a frame are static per (config, input shape), so the host records the sequence of toc3d_* calls ONCE:
 *   toc3d_plan_create(&p); toc3d_plan_begin(p);
 *   ... any toc3d_* entry points, passing toc3d_plan_lane_stream(lane) as `stream`: nothing runs, every kernel launch
 *       (function, grid, block, LDS bytes, argument values) is stored in the plan ...
 *   toc3d_plan_wait(p, a, b): lane a's next launch runs after everything recorded so far on lane b (fork / join of concurrent lanes);
 *   toc3d_plan_end(p, mode);
 * and then enqueues the whole frame with ONE call per frame, toc3d_plan_run(p, stream):
 *   mode 0: hipLaunchKernel on one HIP stream per lane (lane 0 = `stream`, other lanes = streams owned by the plan), cross-lane
 *           edges as hipEvent record / wait; lanes other than 0 start behind the work already on `stream` and are joined into it
 *           at the end, so `stream` orders the frame against its neighbours;
 *   mode 1: one explicitly constructed hipGraph (hipGraphAddKernelNode; lane order and cross-lane edges become node dependencies --
 *           no stream capture), launched on `stream`.
 * Recording is per host thread (begin .. end on one thread; other threads keep launching normally).  Buffers named by the recorded
 * calls are the caller's and must stay allocated and in place while the plan lives; a plan belongs to the device that was current at
 * toc3d_plan_end and must not be run concurrently with itself.  Up to 16 lanes.  Host-side arguments that a call reads on the host
 * (e.g. position_range of toc3d_head_frustum_inputs) are consumed at record time. */
typedef void* toc3d_plan_t;
int toc3d_plan_create(toc3d_plan_t* plan);
int toc3d_plan_destroy(toc3d_plan_t plan);
toc3d_stream_t toc3d_plan_lane_stream(int64_t lane);
int toc3d_plan_begin(toc3d_plan_t plan);
int toc3d_plan_wait(toc3d_plan_t plan, int64_t waiting_lane, int64_t on_lane);
int toc3d_plan_end(toc3d_plan_t plan, int mode);
int64_t toc3d_plan_num_launches(toc3d_plan_t plan);
int toc3d_plan_run(toc3d_plan_t plan, toc3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TOC3D_H_ */
