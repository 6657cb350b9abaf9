// residual epilogues: instantiations of the GEMM kernels of gemm_kernels.h (own translation unit so that the groups build in parallel).
#include "gemm_kernels.h"

int toc3d_gemm_launch_residual(int is_bf16, int epi, int variant, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case TOC3D_EPI_RESIDUAL: return is_bf16 ? launch_epi<bf16_t, TOC3D_EPI_RESIDUAL>(variant, a, s) : launch_epi<float, TOC3D_EPI_RESIDUAL>(variant, a, s);
        case TOC3D_EPI_RESIDUAL_LN: return is_bf16 ? launch_epi<bf16_t, TOC3D_EPI_RESIDUAL_LN>(variant, a, s) : launch_epi<float, TOC3D_EPI_RESIDUAL_LN>(variant, a, s);
        case TOC3D_EPI_RESIDUAL_STATS: return is_bf16 ? launch_epi<bf16_t, TOC3D_EPI_RESIDUAL_STATS>(variant, a, s) : launch_epi<float, TOC3D_EPI_RESIDUAL_STATS>(variant, a, s);
        default: return TOC3D_ERR_ARG;
    }
}
