// Deterministic split-K forms of the residual epilogues (attn.proj eva_vit.py:115,262 / toc3d_eva_vit.py:514,379; mlp.w3 eva_vit.py:49,263 /
// toc3d_eva_vit.py:384): instantiations of gemm_kernels.h with SK = 1 on a few tile variants (own translation unit so that the groups build in parallel).
#include "gemm_kernels.h"

int toc3d_gemm_splitk_tile_dims(int variant) { return sk_tile_dims(variant); }

template <int EPI>
static int launch_one(int dtype, int variant, const GemmArgs& a, hipStream_t s) {
    switch (dtype) {
        case TOC3D_BF16: return launch_epi_sk<bf16_t, EPI>(variant, a, s);
        case TOC3D_F32: return launch_epi_sk<float, EPI>(variant, a, s);
        case TOC3D_F32X3: return launch_epi_sk<float, EPI, 3>(variant, a, s);
        default: return TOC3D_ERR_ARG;
    }
}

int toc3d_gemm_launch_splitk(int dtype, int epi, int variant, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case TOC3D_EPI_RESIDUAL: return launch_one<TOC3D_EPI_RESIDUAL>(dtype, variant, a, s);
        case TOC3D_EPI_RESIDUAL_LN: return launch_one<TOC3D_EPI_RESIDUAL_LN>(dtype, variant, a, s);
        case TOC3D_EPI_RESIDUAL_STATS: return launch_one<TOC3D_EPI_RESIDUAL_STATS>(dtype, variant, a, s);
        default: return TOC3D_ERR_ARG;
    }
}
