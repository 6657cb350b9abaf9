// SwiGLU epilogues: instantiations of the GEMM kernels of gemm_kernels.h (own translation unit so that the groups build in parallel).
#include "gemm_kernels.h"

int toc3d_gemm_launch_swiglu(int is_bf16, int epi, int variant, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case TOC3D_EPI_SWIGLU: return is_bf16 ? launch_epi<bf16_t, TOC3D_EPI_SWIGLU>(variant, a, s) : launch_epi<float, TOC3D_EPI_SWIGLU>(variant, a, s);
        case TOC3D_EPI_SWIGLU_STATS: return is_bf16 ? launch_epi<bf16_t, TOC3D_EPI_SWIGLU_STATS>(variant, a, s) : launch_epi<float, TOC3D_EPI_SWIGLU_STATS>(variant, a, s);
        case TOC3D_EPI_SWIGLU_STATS_LN: return is_bf16 ? launch_epi<bf16_t, TOC3D_EPI_SWIGLU_STATS_LN>(variant, a, s) : launch_epi<float, TOC3D_EPI_SWIGLU_STATS_LN>(variant, a, s);
        default: return TOC3D_ERR_ARG;
    }
}
