// "bf16 x 3" products on f32 operands (TOC3D_DTYPE_F32X3): instantiations of the GEMM kernels of gemm_kernels.h for the epilogues the parity-grade
// paths launch (own translation unit so that the groups build in parallel).
#include "gemm_kernels.h"

int toc3d_gemm_launch_x3(int epi, int variant, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case TOC3D_EPI_BIAS: return launch_epi_x<TOC3D_EPI_BIAS, 3>(variant, a, s);
        case TOC3D_EPI_GELU: return launch_epi_x<TOC3D_EPI_GELU, 3>(variant, a, s);
        case TOC3D_EPI_RESIDUAL: return launch_epi_x<TOC3D_EPI_RESIDUAL, 3>(variant, a, s);
        case TOC3D_EPI_SWIGLU: return launch_epi_x<TOC3D_EPI_SWIGLU, 3>(variant, a, s);
        case TOC3D_EPI_CONV3X3: return launch_epi_x<TOC3D_EPI_CONV3X3, 3>(variant, a, s);
        // SwiGLU.ffn_ln folded across the w1|w2 -> w3 boundary on the parity-grade fast path as well (f32 statistics, f32 row table): no LayerNorm pass over the hidden units
        case TOC3D_EPI_SWIGLU_STATS: return launch_epi_x<TOC3D_EPI_SWIGLU_STATS, 3>(variant, a, s);
        case TOC3D_EPI_RESIDUAL_LN: return launch_epi_x<TOC3D_EPI_RESIDUAL_LN, 3>(variant, a, s);
        // ... and norm2 across the projection -> w1|w2 boundary (round 4): the projection leaves the updated rows as an f32 copy with their statistics
        case TOC3D_EPI_RESIDUAL_STATS: return launch_epi_x<TOC3D_EPI_RESIDUAL_STATS, 3>(variant, a, s);
        case TOC3D_EPI_SWIGLU_STATS_LN: return launch_epi_x<TOC3D_EPI_SWIGLU_STATS_LN, 3>(variant, a, s);
        // ... and the rotating q|k|v epilogue (round 6): RoPE + q scale on the f32 accumulators, rows written as (hi, lo) planes for toc3d_window_attention_rot
        case TOC3D_EPI_QKV_ROPE: return launch_epi_x<TOC3D_EPI_QKV_ROPE, 3>(variant, a, s);
        default: return TOC3D_ERR_ARG;
    }
}
