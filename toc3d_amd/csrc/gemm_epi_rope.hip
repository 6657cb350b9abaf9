// q|k|v projection with RoPE + q scale in the epilogue (bf16): instantiations of the GEMM kernels of gemm_kernels.h (own translation unit so that the groups build in parallel).
#include "gemm_kernels.h"

int toc3d_gemm_launch_rope(int is_bf16, int epi, int variant, const GemmArgs& a, hipStream_t s) {
    if (!is_bf16 || epi != TOC3D_EPI_QKV_ROPE) return TOC3D_ERR_ARG;
    return launch_epi<bf16_t, TOC3D_EPI_QKV_ROPE>(variant, a, s);
}
