// self-normalising epilogues (LayerNorm statistics taken by the GEMM's own K loop) and the plain residual + act-copy producer in front of them:
// instantiations of the GEMM kernels of gemm_kernels.h (own translation unit so that the groups build in parallel).  bf16 only.
#include "gemm_kernels.h"

int toc3d_gemm_launch_lnself(int epi, int variant, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case TOC3D_EPI_RESIDUAL_ACT: return launch_epi<bf16_t, TOC3D_EPI_RESIDUAL_ACT>(variant, a, s);
        case TOC3D_EPI_SWIGLU_LNSELF: return launch_epi<bf16_t, TOC3D_EPI_SWIGLU_LNSELF>(variant, a, s);
        case TOC3D_EPI_RESIDUAL_LNSELF: return launch_epi<bf16_t, TOC3D_EPI_RESIDUAL_LNSELF>(variant, a, s);
        case TOC3D_EPI_QKV_ROPE_LNSELF: return launch_epi<bf16_t, TOC3D_EPI_QKV_ROPE_LNSELF>(variant, a, s);
        default: return TOC3D_ERR_ARG;
    }
}
