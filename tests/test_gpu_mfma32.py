"""GPU: the 32x32x16-MFMA form of the GEMM K loop (tile variants 70-83, bf16; csrc/gemm_kernels.h `lds_frag32`, `block32_to_tiles16`).

The matrix core sums 16 products per instruction instead of 32, so these variants form their own bit class: identical among themselves (every tile
shape walks K in the same order), equal to the 16x16x32 variants up to f32 accumulation order -- checked against an f64 reference of the same bf16
operands (eva_vit.py:44-51,97-99,115 are all `nn.Linear`), for every epilogue the frame launches.
"""
import pytest
import torch

from test_gpu_ops import DEV, S, as_act, pack, relerr, rnd, ru
from toc3d_amd import lib

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not lib.experimental(), reason="measured, not faster (profiles/r04_mfma32_variants.txt): `make EXPERIMENTAL=1`, TOC3D_LIB=libtoc3d_gfx950_exp.so")]
MF32 = (70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83)
dt, tdt = lib.BF16, torch.bfloat16


def ulp_diff(a, b):
    """Fraction of bf16 elements that differ, and the largest difference in units of the larger magnitude's bf16 ulp."""
    a32, b32 = a.float(), b.float()
    d = (a32 - b32).abs()
    ulp = torch.maximum(a32.abs(), b32.abs()).clamp_min(1e-30) * 2.0 ** -7
    return (d > 0).float().mean().item(), (d / ulp).max().item()


@pytest.mark.parametrize("M,N,K", [(777, 640, 512), (6000, 1024, 768), (37, 3072, 1024), (300, 192, 2752)])
def test_mfma32_variants_bias_gelu_residual(M, N, K):
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    a_d, w_d = as_act(A, tdt), pack(W, dt, tdt)
    exact = A.to(tdt).double() @ W.to(tdt).double().T + b.double()
    res = rnd(M, N, seed=4).to(DEV)
    base = {}
    for epi in (lib.EPI_BIAS, lib.EPI_GELU, lib.EPI_RESIDUAL):
        ref = None
        for v in (16,) + MF32:
            f32_out = epi == lib.EPI_RESIDUAL
            out = torch.zeros(M, N, dtype=torch.float32 if f32_out else tdt, device=DEV)
            lib.call("toc3d_linear_ex", dt, epi, v, a_d, K, w_d, K, b.to(DEV), out, N, res if f32_out else None, N if f32_out else 0, 0, None, None, M, N, K, 0, S())
            if v == 16:
                base[epi] = out.clone()
                continue
            if ref is None:
                ref = out.clone()
                if epi == lib.EPI_BIAS:
                    assert relerr(out.float(), exact) < 6e-3
                    frac, worst = ulp_diff(out, base[epi])
                    assert frac < 0.02 and worst <= 1.01, (frac, worst)          # same values up to one rounding step of the bf16 output
                elif epi == lib.EPI_RESIDUAL:
                    assert relerr(out, res.double().cpu() + exact) < 2e-6 * K ** 0.5
                    assert relerr(out, base[epi]) < 2e-6
                else:
                    assert relerr(out.float(), torch.nn.functional.gelu(exact)) < 6e-3
            assert torch.equal(out, ref), f"epilogue {epi}: variant {v} differs from variant {MF32[0]}"


def test_mfma32_variants_carry_the_folded_layernorm_epilogues():
    """proj (+ residual, statistics) -> w1|w2 (norm2 folded, SwiGLU, statistics) -> w3 (ffn_ln folded, + residual): the block half as the bf16 path
    launches it (toc3d_eva_vit.py:366-386), every launch on a 32x32-MFMA variant, against the f64 evaluation on the same rounded intermediates."""
    M, C, Hd = 777, 384, 300
    Hp = ru(Hd, 64)
    eps = 1e-6
    att = as_act(rnd(M, C, seed=1), tdt)
    Wp, bp = rnd(C, C, seed=2, scale=C ** -0.5), rnd(C, seed=3)
    x0 = rnd(M, C, seed=4).to(DEV)
    g2, be2 = (1.0 + 0.3 * rnd(C, seed=5)).to(DEV), (0.2 * rnd(C, seed=6)).to(DEV)
    W1, W2 = rnd(Hd, C, seed=7, scale=C ** -0.5).to(DEV), rnd(Hd, C, seed=8, scale=C ** -0.5).to(DEV)
    b1, b2 = rnd(Hd, seed=9).to(DEV), rnd(Hd, seed=10).to(DEV)
    gf, bf = (1.0 + 0.3 * rnd(Hd, seed=11)).to(DEV), (0.2 * rnd(Hd, seed=12)).to(DEV)
    W3, b3 = rnd(C, Hd, seed=13, scale=Hd ** -0.5).to(DEV), rnd(C, seed=14).to(DEV)
    wp_d = pack(Wp, dt, tdt)
    w12 = torch.empty(2 * Hp, C, dtype=tdt, device=DEV)
    c1_12, c2_12 = torch.empty(2 * Hp, device=DEV), torch.empty(2 * Hp, device=DEV)
    lib.call("toc3d_pack_swiglu_lnfold", dt, W1, W2, b1, b2, g2, be2, Hd, C, w12, c1_12, c2_12, Hp, C, S())
    w3f = torch.empty(ru(C, 128), Hp, dtype=tdt, device=DEV)
    c1, c2 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    lib.call("toc3d_pack_weight_lnfold", dt, W3.contiguous(), gf, bf, b3, C, Hd, w3f, w3f.shape[0], Hp, c1, c2, S())
    cap2, cap = C // 64, ru(-(-2 * Hp // 128), 2)

    def half(vp, v12, v3):
        x = x0.clone()
        a = torch.zeros(M, C, dtype=tdt, device=DEV)
        st2 = torch.zeros(4 + M * cap2 * 2, device=DEV)
        st = torch.zeros(4 + M * cap * 2, device=DEV)
        hid = torch.zeros(M, Hp, dtype=tdt, device=DEV)
        lib.call("toc3d_linear_fused", dt, lib.EPI_RESIDUAL_STATS, vp, att, C, wp_d, C, bp.to(DEV), x, C, x, C, 0, None, None, M, C, C, 0,
                 st2, cap2, None, 0, None, 0, 0.0, a, C, None, S())
        x1 = x.clone()
        lib.call("toc3d_linear_fused", dt, lib.EPI_SWIGLU_STATS_LN, v12, a, C, w12, C, c2_12, hid, Hp, None, 0, 0, None, None, M, 2 * Hp, C, Hd,
                 st, cap, st2, cap2 | (C // 64) << 32, c1_12, C, eps, None, 0, None, S())
        lib.call("toc3d_linear_fused", dt, lib.EPI_RESIDUAL_LN, v3, hid, Hp, w3f, Hp, c2, x, C, x, C, 0, None, None, M, C, Hp, 0,
                 None, 0, st, cap | (-(-2 * Hp // 128)) << 32, c1, Hd, eps, None, 0, None, S())
        return x1, a, hid, x, st2, st

    x1, a, hid, x2, st2, st = half(70, 70, 70)
    # f64 on the rounded intermediates the launches really consumed
    ref1 = x0.double() + att.double() @ Wp.to(tdt).double().T.to(DEV) + bp.double().to(DEV)
    assert relerr(x1, ref1) < 3e-6
    ad = a.double()
    ln2 = (ad - ad.mean(1, keepdim=True)) / torch.sqrt(ad.var(1, unbiased=False, keepdim=True) + eps) * g2.double() + be2.double()
    hr = torch.nn.functional.silu(ln2 @ W1.double().T + b1.double()) * (ln2 @ W2.double().T + b2.double())
    assert relerr(hid[:, :Hd].float(), hr) < 1.2e-2
    hd = hid[:, :Hd].double()
    lnf = (hd - hd.mean(1, keepdim=True)) / torch.sqrt(hd.var(1, unbiased=False, keepdim=True) + eps) * gf.double() + bf.double()
    ref2 = x1.double() + lnf @ W3.double().T + b3.double()
    assert relerr(x2, ref2) < 4e-3
    b1_, a_b, hid_b, x2_b, _, _ = half(16, 16, 16)              # the 16x16x32 variants on the same inputs: same values up to accumulation order / one bf16 rounding step
    assert relerr(x1, b1_) < 2e-6 and ulp_diff(a, a_b)[1] <= 1.01 and relerr(x2, x2_b) < 2e-3
    for vs in ((71, 73, 72), (74, 76, 79), (75, 77, 81), (79, 80, 83), (82, 71, 82), (81, 70, 78)):      # (192-wide tiles: w3 only -- not whole statistics slots)
        o = half(*vs)
        for got, want, what in zip(o, (x1, a, hid, x2, st2, st), ("proj + residual", "bf16 copy", "hidden units", "w3 + residual", "norm2 statistics", "ffn_ln statistics")):
            assert torch.equal(got, want), f"variants {vs}: {what} depends on the 32x32 tile variant"


def test_mfma32_qkv_rope_epilogue():
    from test_gpu_attn_rot import compact_tables, rc_of
    from toc3d_amd import synth
    M, C, L = 600, 256, 16
    a_d = as_act(rnd(M, C, seed=1), tdt)
    w_d, b = pack(rnd(3 * C, C, seed=2, scale=C ** -0.5), dt, tdt), rnd(3 * C, seed=3).to(DEV)
    tab, _ = compact_tables(*synth.rope_tables(L))
    rc = rc_of(torch.arange(M) % (L * L), L)
    outs = {}
    for v in (16, 70, 71, 72, 74, 79):
        o = torch.zeros(M, 3 * C, dtype=tdt, device=DEV)
        lib.call("toc3d_linear_qkv_rope", dt, v, a_d, C, w_d, C, b, o, 3 * C, M, 3 * C, C, rc, tab, L, 0.125, S())
        outs[v] = o
    frac, worst = ulp_diff(outs[70], outs[16])
    assert frac < 0.02 and worst <= 1.01, (frac, worst)
    for v in (71, 72, 74, 79):
        assert torch.equal(outs[v], outs[70]), v
