// bias / GELU / implicit 3x3 conv epilogues: instantiations of the GEMM kernels of gemm_kernels.h (own translation unit so that the groups build in parallel).
#include "gemm_kernels.h"

int toc3d_gemm_launch_plain(int is_bf16, int epi, int variant, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case TOC3D_EPI_BIAS: return is_bf16 ? launch_epi<bf16_t, TOC3D_EPI_BIAS>(variant, a, s) : launch_epi<float, TOC3D_EPI_BIAS>(variant, a, s);
        case TOC3D_EPI_GELU: return is_bf16 ? launch_epi<bf16_t, TOC3D_EPI_GELU>(variant, a, s) : launch_epi<float, TOC3D_EPI_GELU>(variant, a, s);
        case TOC3D_EPI_CONV3X3: return is_bf16 ? launch_epi<bf16_t, TOC3D_EPI_CONV3X3>(variant, a, s) : launch_epi<float, TOC3D_EPI_CONV3X3>(variant, a, s);
        default: return TOC3D_ERR_ARG;
    }
}
