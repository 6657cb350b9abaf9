"""GPU: LayerNorms folded into the consuming GEMM with the row statistics taken by that GEMM's own K loop (EPI_SWIGLU_LNSELF, EPI_RESIDUAL_LNSELF,
EPI_QKV_ROPE_LNSELF; producer EPI_RESIDUAL_ACT) -- norm2 / ffn_ln / norm1 of eva_vit.py:258-263,44-51 without a LayerNorm launch or a statistics
buffer.  Checked against f64 references on the same rounded operands, against the explicit-LayerNorm sequences, and for tile-variant independence."""
import pytest
import torch

from toc3d_amd import lib, synth

from test_gpu_ops import DEV, S, as_act, pack, relerr, rnd, ru
from test_gpu_attn_rot import compact_tables, rc_of, rope_ref

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not lib.experimental(), reason="round-3 experiment: `make EXPERIMENTAL=1`, TOC3D_LIB=libtoc3d_gfx950_exp.so")]
BF, TBF = lib.BF16, torch.bfloat16
EPS = 1e-6


def fused(epi, v, A, lda, W, ldw, bias, out, ldo, res, ldr, rep, rep_index, M, N, K, nv, extra):
    lib.call("toc3d_linear_fused", BF, epi, v, A, lda, W, ldw, bias, out, ldo, res, ldr, 0, rep, rep_index, M, N, K, nv, *extra, S())


@pytest.mark.parametrize("M,C,Hd", [(777, 384, 300), (1300, 1024, 2730)])
def test_norm2_and_ffn_ln_statistics_from_the_k_loop(M, C, Hd):
    Hp = ru(Hd, 64)
    att = as_act(rnd(M, C, seed=1), TBF)
    Wp, bp = rnd(C, C, seed=2, scale=C ** -0.5), rnd(C, seed=3).to(DEV)
    wproj = pack(Wp, BF, TBF)
    x0 = (3.0 * rnd(M, C, seed=4) + 0.7).to(DEV)                      # residual stream with a non-zero mean
    g2, b2 = (1.0 + 0.3 * rnd(C, seed=5)).to(DEV), (0.2 * rnd(C, seed=6)).to(DEV)
    w1, w2 = rnd(Hd, C, seed=7, scale=C ** -0.5).to(DEV), rnd(Hd, C, seed=8, scale=C ** -0.5).to(DEV)
    bb1, bb2 = rnd(Hd, seed=9).to(DEV), rnd(Hd, seed=10).to(DEV)
    gf, bf = (1.0 + 0.3 * rnd(Hd, seed=11)).to(DEV), (0.2 * rnd(Hd, seed=12)).to(DEV)
    W3, b3 = rnd(C, Hd, seed=13, scale=Hd ** -0.5).to(DEV), rnd(C, seed=14).to(DEV)
    rep_index = torch.full((M,), -1, dtype=torch.int32, device=DEV)
    rep_index[::40] = torch.arange(len(range(0, M, 40)), dtype=torch.int32, device=DEV)
    nrep = int((rep_index >= 0).sum())
    none10 = lib.NO_FUSED

    # ---- producer: EPI_RESIDUAL_ACT = EPI_RESIDUAL + the rounded copy ----
    x_ref = x0.clone()
    rep0 = torch.zeros(nrep, C, device=DEV)
    lib.call("toc3d_linear_ex", BF, lib.EPI_RESIDUAL, 16, att, C, wproj, C, bp, x_ref, C, x_ref, C, 0, rep0, rep_index, M, C, C, 0, S())
    for v in (1, 8, 14, 16, 17, 19, 26, 29, 49, 114, 117, 126):
        x = x0.clone()
        a_raw = torch.full((M, C), 9.0, dtype=TBF, device=DEV)
        rep = torch.zeros(nrep, C, device=DEV)
        fused(lib.EPI_RESIDUAL_ACT, v, att, C, wproj, C, bp, x, C, x, C, rep, rep_index, M, C, C, 0, none10[:7] + (a_raw, C, None))
        assert torch.equal(x, x_ref) and torch.equal(rep, rep0) and torch.equal(a_raw, x_ref.to(TBF)), f"variant {v}"
    a_raw = x_ref.to(TBF)

    # ---- norm2 inside w1|w2 ----
    w12f = torch.empty(2 * Hp, C, dtype=TBF, device=DEV)
    c1_12, c2_12 = torch.empty(2 * Hp, device=DEV), torch.empty(2 * Hp, device=DEV)
    lib.call("toc3d_pack_swiglu_lnfold", BF, w1, w2, bb1, bb2, g2, b2, Hd, C, w12f, c1_12, c2_12, Hp, C, S())
    xd = a_raw.double()                                              # the kernel normalises the rounded rows it multiplies
    ln = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + EPS) * g2.double() + b2.double()
    h_ref = torch.nn.functional.silu(ln @ w1.double().T + bb1.double()) * (ln @ w2.double().T + bb2.double())
    # explicit sequence for scale: LayerNorm launch on the f32 stream + plain SwiGLU GEMM
    a_ln = torch.zeros(M, C, dtype=TBF, device=DEV)
    lib.call("toc3d_layernorm_rows", BF, x_ref, C, None, None, g2, b2, EPS, a_ln, C, M, C, S())
    w12 = torch.empty(2 * Hp, C, dtype=TBF, device=DEV)
    b12 = torch.empty(2 * Hp, device=DEV)
    lib.call("toc3d_pack_swiglu", BF, w1, w2, bb1, bb2, Hd, C, w12, b12, Hp, C, S())
    hid_seq = torch.zeros(M, Hp, dtype=TBF, device=DEV)
    lib.call("toc3d_linear_ex", BF, lib.EPI_SWIGLU, 16, a_ln, C, w12, C, b12, hid_seq, Hp, None, 0, 0, None, None, M, 2 * Hp, C, Hd, S())
    ref_h = None
    for v in (1, 8, 10, 15, 16, 17, 19, 22, 24, 26, 28, 29, 47, 49, 51, 116, 117, 126, 149):
        hid = torch.full((M, Hp), 9.0, dtype=TBF, device=DEV)
        fused(lib.EPI_SWIGLU_LNSELF, v, a_raw, C, w12f, C, c2_12, hid, Hp, None, 0, None, None, M, 2 * Hp, C, Hd, (None, 0, None, 0, c1_12, C, EPS, None, 0, None))
        if ref_h is None:
            ref_h = hid.clone()
            e_self, e_seq = relerr(hid[:, :Hd], h_ref), relerr(hid_seq[:, :Hd], h_ref)
            print(f"[norm2 in the K loop, M={M} C={C}] hidden units rel err vs f64: self-normalising {e_self:.3e}, explicit LayerNorm launch {e_seq:.3e}")
            assert e_self < 1.5e-2 and e_self < 1.5 * e_seq + 1e-3
            assert torch.count_nonzero(hid[:, Hd:]) == 0
        assert torch.equal(hid, ref_h), f"variant {v}: self-normalising w1|w2 epilogue depends on the tile variant"

    # ---- ffn_ln inside w3 ----
    hid0 = ref_h
    w3f = torch.zeros(ru(C, 128), Hp, dtype=TBF, device=DEV)
    c1, c2 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    lib.call("toc3d_pack_weight_lnfold", BF, W3.contiguous(), gf, bf, b3, C, Hd, w3f, w3f.shape[0], Hp, c1, c2, S())
    h = hid0[:, :Hd].double()
    lnh = (h - h.mean(1, keepdim=True)) / torch.sqrt(h.var(1, unbiased=False, keepdim=True) + EPS) * gf.double() + bf.double()
    delta_ref = lnh @ W3.double().T + b3.double()
    ref = x_ref.double() + delta_ref
    hln = torch.zeros(M, Hp, dtype=TBF, device=DEV)
    lib.call("toc3d_layernorm_act", BF, hid0, Hp, gf, bf, EPS, hln, Hp, M, Hd, S())
    out_seq = x_ref.clone()
    lib.call("toc3d_linear_ex", BF, lib.EPI_RESIDUAL, 16, hln, Hp, pack(W3.cpu(), BF, TBF), Hp, b3, out_seq, C, out_seq, C, 0, None, None, M, C, Hp, 0, S())
    ref_out = None
    for v in (1, 8, 9, 10, 13, 14, 16, 17, 19, 22, 26, 28, 29, 33, 45, 47, 49, 51, 52, 53, 110, 114, 116, 117, 126, 145, 149, 151, 152):
        for with_copy in (False, True):
            out = x_ref.clone()
            rep = torch.zeros(nrep, C, device=DEV)
            a_next = torch.full((M, C), 9.0, dtype=TBF, device=DEV)
            fused(lib.EPI_RESIDUAL_LNSELF, v, hid0, Hp, w3f, Hp, c2, out, C, out, C, rep, rep_index, M, C, Hp, 0,
                  (None, 0, None, 0, c1, Hd, EPS, a_next if with_copy else None, C if with_copy else 0, None))
            if ref_out is None:
                ref_out, ref_rep = out.clone(), rep.clone()
                e_self, e_seq = relerr(out, ref), relerr(out_seq, ref)
                print(f"[ffn_ln in the K loop, M={M}] rel err vs f64: self-normalising {e_self:.3e}, explicit LayerNorm pass {e_seq:.3e}")
                assert e_self < 4e-3 and e_self < 1.5 * e_seq + 1e-4
                assert relerr(rep, delta_ref[rep_index.cpu() >= 0]) < 6e-3
            assert torch.equal(out, ref_out) and torch.equal(rep, ref_rep), f"variant {v}: self-normalising w3 epilogue depends on the tile variant"
            if with_copy:
                assert torch.equal(a_next, ref_out.to(TBF))
    with pytest.raises(RuntimeError, match="cannot serve"):
        fused(lib.EPI_SWIGLU_LNSELF, 60, a_raw, C, w12f, C, c2_12, hid, Hp, None, 0, None, None, M, 2 * Hp, C, Hd, (None, 0, None, 0, c1_12, C, EPS, None, 0, None))
    with pytest.raises(RuntimeError, match="col_sums"):
        fused(lib.EPI_SWIGLU_LNSELF, 16, a_raw, C, w12f, C, c2_12, hid, Hp, None, 0, None, None, M, 2 * Hp, C, Hd, (None, 0, None, 0, None, C, EPS, None, 0, None))


@pytest.mark.parametrize("C,M,L", [(128, 333, 16), (1024, 777, 20)])
def test_norm1_statistics_from_the_qkv_k_loop(C, M, L):
    heads = C // 64
    cos, sin = synth.rope_tables(L)
    x = 2.0 * rnd(M, C, seed=1) + 0.5
    W, b = rnd(3 * C, C, seed=2, scale=C ** -0.5), rnd(3 * C, seed=3)
    g1, b1 = 1.0 + 0.3 * rnd(C, seed=5), 0.2 * rnd(C, seed=6)
    gen = torch.Generator().manual_seed(4)
    slots = torch.randint(0, L * L, (M,), generator=gen)
    a_raw = as_act(x, TBF)
    xd = a_raw.double().cpu()
    ln = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + EPS) * g1.double() + b1.double()
    y = (ln @ W.double().T + b.double()).view(M, 3, heads, 64)
    cs, sn = cos[slots].double()[:, None, :], sin[slots].double()[:, None, :]
    ref = torch.stack([rope_ref(y[:, 0], cs, sn) * 64 ** -0.5, rope_ref(y[:, 1], cs, sn), y[:, 2]], 1).reshape(M, 3 * C)
    tab, _ = compact_tables(cos, sin)
    wf = torch.zeros(ru(3 * C, 128), C, dtype=TBF, device=DEV)
    c1, c2 = torch.empty(3 * C, device=DEV), torch.empty(3 * C, device=DEV)
    lib.call("toc3d_pack_weight_lnfold", BF, W.to(DEV).contiguous(), g1.to(DEV), b1.to(DEV), b.to(DEV), 3 * C, C, wf, wf.shape[0], C, c1, c2, S())
    rc = rc_of(slots, L)
    first = None
    for v in (0, 1, 8, 14, 16, 17, 19, 29, 45, 49, 52, 53, 116, 117, 149, 152):
        out = torch.empty(M, 3 * C, dtype=TBF, device=DEV)
        lib.call("toc3d_linear_qkv_rope_ln", BF, v, a_raw, C, wf, C, c2, out, 3 * C, M, 3 * C, C, rc, tab, L, 64 ** -0.5, c1, C, EPS, S())
        if first is None:
            first = out.clone()
            e = relerr(out.float(), ref)
            print(f"[norm1 in the q|k|v K loop, C={C}] rel err vs f64 {e:.3e}")
            assert e < 1.2e-2
        assert torch.equal(out, first), f"variant {v} differs"
