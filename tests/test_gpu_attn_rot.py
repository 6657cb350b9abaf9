"""GPU: the pre-rotated attention path of the bf16 build (round 3) against the oracle through the C ABI:
toc3d_linear_qkv_rope (RoPE + q scale in the q|k|v projection's epilogue) and toc3d_window_attention_rot (K / V staged by DMA,
V read with the transposing LDS read).  Tolerances as in test_gpu_ops.py: the reference sees the same bf16-rounded operands."""
import pytest
import torch

from oracle import toc3d_oracle as O
from toc3d_amd import configs, lib, synth

from test_gpu_ops import DEV, S, as_act, pack, relerr, rnd

pytestmark = pytest.mark.gpu
BF, TBF = lib.BF16, torch.bfloat16
F32T = torch.float32
# the two precisions of the pre-rotated path: bf16 buffers, or (round 6) f32-sized buffers holding (hi, lo) bf16 planes with bf16 x 3 products (precision "fp32x3")
PRECS = ["bf16", "x3"]


def planes_decode(t):
    """A (hi, lo) planes buffer (f32-typed tensor [rows, ld], ld % 32 == 0; include/toc3d.h TOC3D_DTYPE_F32X3P) -> the f32 values hi + lo."""
    rows, ld = t.shape
    b = t.contiguous().view(torch.bfloat16).view(rows, ld // 32, 2, 32).float()
    return (b[:, :, 0] + b[:, :, 1]).reshape(rows, ld)


def planes_encode(x):
    """f32 device tensor [rows, ld] -> planes, through the library's own converter."""
    x = x.contiguous().clone()
    lib.call("toc3d_x3_planes", x, x.shape[1], x, x.shape[1], x.shape[0], x.shape[1], S())
    return x


def pack_p(w, prec):
    return pack(w, BF, TBF) if prec == "bf16" else planes_encode(pack(w, lib.F32, F32T))


def act_p(x, prec, planes=True):
    if prec == "bf16":
        return as_act(x, TBF)
    a = as_act(x, F32T)
    return planes_encode(a) if planes else a


def values(t, prec):
    return t.float() if prec == "bf16" else planes_decode(t)


def compact_tables(cos, sin):
    """[L*L, 64] reference buffers -> the [2, L, 16] tables toc3d_linear_qkv_rope consumes (what backbone._pack_blocks builds)."""
    L = int(round(cos.shape[0] ** 0.5))
    out = []
    for t in (cos, sin):
        g = t.view(L, L, 64)
        out.append(torch.stack([g[:, 0, 0:32:2], g[0, :, 32:64:2]]))
    return torch.stack(out).contiguous().to(DEV), L


def rc_of(slots, L):
    return (((slots // L) << 16) | (slots % L)).to(torch.int32).contiguous().to(DEV)


def rope_ref(x, cos, sin):
    """eva_utils.py:378-379 with rotate_half over pairs (2t, 2t+1); x [..., 64], cos / sin broadcastable."""
    x2 = torch.stack([-x[..., 1::2], x[..., 0::2]], -1).flatten(-2)
    return x * cos + x2 * sin


def qkv_rope(a_d, wqkv_p, bqkv, M, C, rc, tab, L, variant=0, prec="bf16", a_planes=True):
    out = torch.empty(M, 3 * C, dtype=TBF if prec == "bf16" else F32T, device=DEV)
    dt = BF if prec == "bf16" else (lib.F32X3P if a_planes else lib.F32X3WO)
    lib.call("toc3d_linear_qkv_rope", dt, variant, a_d, C, wqkv_p, C, bqkv, out, 3 * C, M, 3 * C, C, rc, tab, L, lib.ATTN_ROT_Q_SCALE, S())
    return out


def attn_rot(prec, qkv, C, out, rows, slots, count, count_k, npad, pad_rot, stride, nW, max_count, heads, v_bias):
    lib.call("toc3d_window_attention_rot", BF if prec == "bf16" else lib.F32X3P, qkv, 3 * C, out, C, rows, slots, count, count_k, npad, pad_rot, stride, nW, max_count, heads,
             v_bias, 0, None, None, 0, S())


def test_qkv_projection_with_rope_in_the_epilogue_on_planes():
    """bf16 x 3 form of the rotating epilogue: A (planes or plain f32) . W (planes), RoPE + q scale on the f32 accumulators, rows written as (hi, lo) planes.  Against f64
    on the f32 operands: product error <= ~2^-16, the planes carry 16 mantissa bits."""
    C, M, L = 1024, 777, 20
    heads = C // 64
    cos, sin = synth.rope_tables(L)
    A, W, b = rnd(M, C, seed=1), rnd(3 * C, C, seed=2, scale=C ** -0.5), rnd(3 * C, seed=3)
    g = torch.Generator().manual_seed(4)
    slots = torch.randint(0, L * L, (M,), generator=g)
    y = (A.double() @ W.double().T + b.double()).view(M, 3, heads, 64)
    cs, sn = cos[slots].double()[:, None, :], sin[slots].double()[:, None, :]
    ref = torch.stack([rope_ref(y[:, 0], cs, sn) * lib.ATTN_ROT_Q_SCALE, rope_ref(y[:, 1], cs, sn), y[:, 2]], 1).reshape(M, 3 * C)
    tab, _ = compact_tables(cos, sin)
    w_d = pack_p(W, "x3")
    out = qkv_rope(act_p(A, "x3"), w_d, b.to(DEV), M, C, rc_of(slots, L), tab, L, prec="x3")
    assert relerr(planes_decode(out), ref) < 5e-5
    out_plain_a = qkv_rope(act_p(A, "x3", planes=False), w_d, b.to(DEV), M, C, rc_of(slots, L), tab, L, prec="x3", a_planes=False)
    assert torch.equal(out_plain_a.view(torch.int32), out.view(torch.int32)), "A split in the kernel or delivered as planes: the same bits"
    for v in (1, 8, 9, 10, 14, 16, 17, 19, 22, 26, 28, 29, 33, 45, 47, 49, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 116, 117, 149, 152, 155, 160, 161):
        assert torch.equal(qkv_rope(act_p(A, "x3"), w_d, b.to(DEV), M, C, rc_of(slots, L), tab, L, v, prec="x3").view(torch.int32), out.view(torch.int32)), f"variant {v} differs"


@pytest.mark.parametrize("C,M,L", [(128, 333, 16), (1024, 777, 20)])
def test_qkv_projection_with_rope_in_the_epilogue(C, M, L):
    heads = C // 64
    cos, sin = synth.rope_tables(L)
    A, W, b = rnd(M, C, seed=1), rnd(3 * C, C, seed=2, scale=C ** -0.5), rnd(3 * C, seed=3)
    g = torch.Generator().manual_seed(4)
    slots = torch.randint(0, L * L, (M,), generator=g)
    y = (A.to(TBF).double() @ W.to(TBF).double().T + b.double()).view(M, 3, heads, 64)
    cs, sn = cos[slots].double()[:, None, :], sin[slots].double()[:, None, :]
    ref = torch.stack([rope_ref(y[:, 0], cs, sn) * lib.ATTN_ROT_Q_SCALE, rope_ref(y[:, 1], cs, sn), y[:, 2]], 1).reshape(M, 3 * C)      # (head_dim^-0.5 * log2(e): the attention's softmax is exp2-based)
    tab, _ = compact_tables(cos, sin)
    a_d, w_d = as_act(A, TBF), pack(W, BF, TBF)
    out = qkv_rope(a_d, w_d, b.to(DEV), M, C, rc_of(slots, L), tab, L)
    assert relerr(out.float(), ref) < 6e-3
    # one rounding: every element within half a bf16 ulp (+ accumulation order) of the exact value
    assert ((out.float().cpu().double() - ref).abs() <= ref.abs() * 2.0 ** -8 + 1e-6).all()
    plain = torch.empty(M, 3 * C, dtype=TBF, device=DEV)
    lib.call("toc3d_linear", BF, lib.EPI_BIAS, a_d, C, w_d, C, b.to(DEV), plain, 3 * C, None, 0, 0, None, None, M, 3 * C, C, 0, S())
    assert torch.equal(out[:, 2 * C:], plain[:, 2 * C:]), "the v columns are the plain projection"
    for v in (1, 8, 14, 16, 17, 19, 29, 45, 49, 52, 53, 54, 55, 56, 57, 58, 60, 61, 62, 63, 64, 65, 66, 116, 117, 149, 152, 154, 160, 161):
        assert torch.equal(qkv_rope(a_d, w_d, b.to(DEV), M, C, rc_of(slots, L), tab, L, v), out), f"variant {v} differs"


def _proj_weights(sd, pre, C, prec="bf16"):
    wqkv = torch.cat([sd[pre + "q_proj.weight"], sd[pre + "k_proj.weight"], sd[pre + "v_proj.weight"]])
    bqkv = torch.cat([sd[pre + "q_bias"], torch.zeros(C), sd[pre + "v_bias"]])
    return pack_p(wqkv, prec), bqkv.to(DEV)


TOL = {"bf16": 3e-2, "x3": 1e-4}               # against the oracle's f32 attention: bf16 operands round at 2^-9; the x3 products at ~2^-16


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("L", [16, 20])
def test_rot_attention_dense_windows_with_analytic_pads(L, prec):
    """Block.forward attention part (eva_vit.py:249-262) on the pre-rotated path: 256 / 400-key windows, ragged edge windows."""
    cfg = configs.get("toc3d_tiny")
    sd = synth.make_state_dict(cfg)
    C, heads, V, h, w = cfg["embed_dim"], cfg["num_heads"], 2, 20, 50
    pre = "blocks.2.attn." if L == 20 else "blocks.0.attn."
    y = rnd(V, h, w, C, seed=7)
    yw, pad_hw = O.window_partition(y, L)
    nB = yw.shape[0]
    sd2 = dict(sd)
    sd2[pre + "proj.weight"], sd2[pre + "proj.bias"] = torch.eye(C), torch.zeros(C)
    cos, sin = sd[pre + "rope.freqs_cos"], sd[pre + "rope.freqs_sin"]
    ref = O.attention(yw.reshape(nB, L * L, C), sd2, pre, heads, cos, sin)
    ref = O.window_unpartition(ref.reshape(nB, L, L, C), L, pad_hw, (h, w)).reshape(-1, C)
    M = V * h * w
    wq, bq = _proj_weights(sd, pre, C, prec)
    tab, _ = compact_tables(cos, sin)
    r = torch.arange(h).view(1, h, 1).expand(V, h, w)
    c = torch.arange(w).view(1, 1, w).expand(V, h, w)
    rc = (((r % L) << 16) | (c % L)).reshape(-1).to(torch.int32).to(DEV)
    qkv = qkv_rope(act_p(y.reshape(M, C), prec), wq, bq, M, C, rc, tab, L, prec=prec)
    nW, N = nB, L * L
    rows = torch.empty(nW, N, dtype=torch.int32, device=DEV)
    slots, count, npad = torch.empty_like(rows), torch.empty(nW, dtype=torch.int32, device=DEV), torch.empty(nW, dtype=torch.int32, device=DEV)
    lib.call("toc3d_window_map_dense", V, h, w, L, rows, slots, count, npad, S())
    out = torch.zeros(M, C, dtype=qkv.dtype, device=DEV)
    attn_rot(prec, qkv, C, out, rows, slots, count, None, npad, None, N, nW, int(count.max()), heads, sd[pre + "v_bias"].to(DEV))
    err = relerr(values(out, prec), ref)
    assert err < TOL[prec], err


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("n", [33, 77, 103, 129, 161, 201, 256, 300, 401])
def test_rot_attention_selected_slots(n, prec):
    """ToC3DEVAAttention (toc3d_eva_vit.py:484-518): compact rows, RoPE rows gathered by slot index -- every instantiation of the kernel."""
    cfg = configs.get("toc3d_tiny")
    sd = synth.make_state_dict(cfg)
    C, heads, nW = cfg["embed_dim"], cfg["num_heads"], 5
    pre = "blocks.5.attn."
    cosT, sinT = synth.rope_tables(21)                          # 441 slots: room for n = 401 distinct ones
    y = rnd(nW, n, C, seed=11)
    g = torch.Generator().manual_seed(12)
    slots = torch.stack([torch.randperm(441, generator=g)[:n] for _ in range(nW)])
    sd2 = dict(sd)
    sd2[pre + "proj.weight"], sd2[pre + "proj.bias"] = torch.eye(C), torch.zeros(C)
    ref = O.attention(y, sd2, pre, heads, cosT[slots], sinT[slots]).reshape(-1, C)
    M = nW * n
    wq, bq = _proj_weights(sd, pre, C, prec)
    tab, L = compact_tables(cosT, sinT)
    qkv = qkv_rope(act_p(y.reshape(M, C), prec), wq, bq, M, C, rc_of(slots.reshape(-1), L), tab, L, prec=prec)
    # scattered compact rows: the kernel must go through the index list
    perm = torch.randperm(M, generator=g)
    qkv_s = torch.empty_like(qkv)
    qkv_s[perm.to(DEV)] = qkv
    rows = perm.to(torch.int32).reshape(nW, n).to(DEV)
    count = torch.full((nW,), n, dtype=torch.int32, device=DEV)
    out = torch.zeros(M, C, dtype=qkv.dtype, device=DEV)
    attn_rot(prec, qkv_s, C, out, rows, None, count, None, None, None, n, nW, n, heads, None)
    err = relerr(values(out[perm.to(DEV)], prec), ref)
    assert err < TOL[prec], err


@pytest.mark.parametrize("prec", PRECS)
def test_rot_attention_virtual_pad_keys_equal_explicit_pad_rows(prec):
    """toc3d_eva_vit.py:414,421,372: kept padded slots are LN(0) = beta rows.  As virtual keys (rows = -1, k taken from pad_rot[slot]) they must give
    the real rows exactly the output they get when the pads are explicit rows; ragged query counts per window."""
    cfg = configs.get("toc3d_tiny")
    sd = synth.make_state_dict(cfg)
    C, heads, nW, n_real, n_pad, L = cfg["embed_dim"], cfg["num_heads"], 4, 37, 60, 16
    pre = "blocks.3.attn."
    tab, _ = compact_tables(sd[pre + "rope.freqs_cos"], sd[pre + "rope.freqs_sin"])
    beta_row = (0.1 * rnd(1, C, seed=21)).to(TBF).float()
    y = torch.cat([rnd(nW, n_real, C, seed=22), beta_row.expand(nW, n_pad, C)], 1)
    n = n_real + n_pad
    gsl = torch.Generator().manual_seed(23)
    slots = torch.stack([torch.randperm(256, generator=gsl)[:n] for _ in range(nW)]).int()
    wq, bq = _proj_weights(sd, pre, C, prec)
    M = nW * n
    qkv = qkv_rope(act_p(y.reshape(M, C), prec), wq, bq, M, C, rc_of(slots.reshape(-1).long(), L), tab, L, prec=prec)
    rows = torch.arange(M, dtype=torch.int32).reshape(nW, n).to(DEV)
    cnt = torch.full((nW,), n, dtype=torch.int32, device=DEV)
    full = torch.zeros(M, C, dtype=qkv.dtype, device=DEV)
    attn_rot(prec, qkv, C, full, rows, slots.to(DEV), cnt, None, None, None, n, nW, n, heads, None)
    # the pad row rotated for every window slot, by the same epilogue (what ToC3DEVAViT._pack builds: on the x3 path from the PLAIN f32 LayerNorm row)
    pad_rot = qkv_rope(act_p(beta_row.expand(256, C).contiguous(), prec, planes=False), wq, bq, 256, C, rc_of(torch.arange(256), L), tab, L, prec=prec, a_planes=False)
    rows_v = rows.clone()
    rows_v[:, n_real:] = -1
    cq = torch.full((nW,), n_real, dtype=torch.int32, device=DEV)
    virt = torch.zeros(M, C, dtype=qkv.dtype, device=DEV)
    attn_rot(prec, qkv, C, virt, rows_v, slots.to(DEV), cq, cnt, None, pad_rot, n, nW, n_real, heads, None)
    fr = full.view(nW, n, C)[:, :n_real].float()
    vr = virt.view(nW, n, C)[:, :n_real].float()
    assert torch.equal(fr, vr)
    assert (virt.view(nW, n, C)[:, n_real:] == 0).all(), "pad rows must not be written"


def test_rot_attention_is_bit_stable_and_rides_prefetch():
    """Many workgroups per CU, repeated launches, with and without the weight-prefetch rows in the grid: identical bits every time."""
    import ctypes
    V, h, w, L, C, heads = 12, 20, 50, 16, 128, 2
    M, N = V * h * w, L * L
    nW = V * 2 * 4
    qkv = as_act(rnd(M, 3 * C, seed=3), TBF)
    rows = torch.empty(nW, N, dtype=torch.int32, device=DEV)
    slots, count, npad = torch.empty_like(rows), torch.empty(nW, dtype=torch.int32, device=DEV), torch.empty(nW, dtype=torch.int32, device=DEV)
    lib.call("toc3d_window_map_dense", V, h, w, L, rows, slots, count, npad, S())
    vb = rnd(C, seed=6).to(DEV)
    junk = torch.randn(1 << 20, device=DEV)
    ptrs = (ctypes.c_void_p * 1)(junk.data_ptr())
    nb = (ctypes.c_int64 * 1)(junk.numel() * 4)
    outs = [torch.zeros(M, C, dtype=TBF, device=DEV) for _ in range(12)]
    for i, o in enumerate(outs):
        lib.call("toc3d_window_attention_rot", BF, qkv, 3 * C, o, C, rows, slots, count, None, npad, None, N, nW, int(count.max()), heads, vb,
                 i % 2, ptrs, nb, 64, S())
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o.view(torch.uint8), outs[0].view(torch.uint8))
    assert bool(torch.isfinite(outs[0].float()).all()) and float(outs[0].float().abs().max()) > 0


@pytest.mark.parametrize("L", [16, 20])
def test_rot_attention_on_planes_is_bit_stable_with_co_resident_workgroups(L):
    """The planes form under load: 12 views (96 / 36 windows x 2 heads, several workgroups per CU; L = 20: the 128-key super-tile kernel with its per-tile barriers and
    restaging), repeated launches beside a GEMM stream on a second HIP stream: identical bits every time (the packed-FP32 erratum and the barrier-fence race of
    LABNOTES.md were both found by this kind of test)."""
    V, h, w, C, heads = 12, 20, 50, 128, 2
    M, N = V * h * w, L * L
    nW = V * (-(-h // L)) * (-(-w // L))
    qkv = planes_encode(rnd(M, 3 * C, seed=3).to(DEV))
    rows = torch.empty(nW, N, dtype=torch.int32, device=DEV)
    slots, count, npad = torch.empty_like(rows), torch.empty(nW, dtype=torch.int32, device=DEV), torch.empty(nW, dtype=torch.int32, device=DEV)
    lib.call("toc3d_window_map_dense", V, h, w, L, rows, slots, count, npad, S())
    vb = rnd(C, seed=6).to(DEV)
    big_a, big_w = as_act(rnd(4096, 1024, seed=24), TBF), pack(rnd(2048, 1024, seed=25, scale=1 / 32), BF, TBF)
    big_o, big_b = torch.zeros(4096, 2048, dtype=TBF, device=DEV), torch.zeros(2048, device=DEV)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    outs = [torch.zeros(M, C, dtype=F32T, device=DEV) for _ in range(12)]
    for i, o in enumerate(outs):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                for _ in range(3):
                    lib.call("toc3d_linear_ex", BF, lib.EPI_BIAS, 17, big_a, 1024, big_w, 1024, big_b, big_o, 2048, None, 0, 0, None, None, 4096, 2048, 1024, 0, S())
        attn_rot("x3", qkv, C, o, rows, slots, count, None, npad, None, N, nW, int(count.max()), heads, vb)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int32), outs[0].view(torch.int32))
    v = planes_decode(outs[0])
    assert bool(torch.isfinite(v).all()) and float(v.abs().max()) > 0


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("n,np_pad", [(129, 0), (201, 0), (256, 0), (96, 160), (400, 0)])
def test_rot_attention_deferred_rescale_is_the_exact_softmax(n, np_pad, prec):
    """The one-pass softmax of round 5 moves a query's reference point only when a 32-key chunk's maximum exceeds it by more than 8 (log2 units): the branch
    that rescales sum and O^T is (almost) never taken after the first chunk on benign data, so it gets its own input (cdna_hip_programming.md rule 26): spikes
    planted in LATER chunks -- one key far above everything before it, a second one above that, both only for some queries -- against an f64 softmax of the
    same bf16 operands; plus the benign case and the analytic zero-pad keys (np_pad > 0: their score 0 is the starting reference point)."""
    heads, nW = 2, 3
    C = heads * 64
    g = torch.Generator().manual_seed(n)
    M = nW * n
    q = torch.randn(M, heads, 64, generator=g) * 0.6
    k = torch.randn(M, heads, 64, generator=g)
    v = torch.randn(M, heads, 64, generator=g)
    # window 0: key 70 spikes for queries 0..39, key n - 5 spikes higher for queries 20..59 (both in later chunks than the first)
    for key, qs, amp in ((70, range(0, 40), 6.0), (n - 5, range(20, 60), 12.0)):
        if key < n:
            k[key] = 0.0
            for qi in qs:
                if qi < n:
                    k[key] += q[qi] / 40.0
            k[key] *= amp * 40.0 / 6.0
    if prec == "x3":
        q, k, v = (t.to(TBF).float() for t in (q, k, v))      # the same bf16-representable operands: hi = the value, lo = 0
    qkv = torch.cat([q.reshape(M, C), k.reshape(M, C), v.reshape(M, C)], 1).to(TBF).to(DEV).contiguous()
    if prec == "x3":
        qkv = planes_encode(qkv.float())
    rows = torch.arange(M, dtype=torch.int32).view(nW, n)
    stride = (n + 15) // 16 * 16
    rows_p = torch.zeros(nW, stride, dtype=torch.int32)
    rows_p[:, :n] = rows
    count = torch.full((nW,), n, dtype=torch.int32)
    npad = torch.full((nW,), np_pad, dtype=torch.int32) if np_pad else None
    vb = (torch.randn(C, generator=g) * 0.3)
    out = torch.zeros(M, C, dtype=qkv.dtype, device=DEV)
    attn_rot(prec, qkv, C, out, rows_p.to(DEV), rows_p.to(DEV), count.to(DEV), None, None if npad is None else npad.to(DEV), None, stride, nW, n, heads,
             vb.to(DEV) if np_pad else None)
    qb, kb, vbf = (t.to(TBF).double().view(nW, n, heads, 64) for t in (q, k, v))
    S_ = torch.einsum("wqhd,wkhd->whqk", qb, kb) * 0.6931471805599453          # the kernel exponentiates with exp2: exp2(s) = exp(s ln 2)
    if np_pad:                                                                    # zero-pad keys: score 0, value = v_bias (attention.hip header)
        S_ = torch.cat([S_, torch.zeros(nW, heads, n, np_pad, dtype=torch.float64)], -1)
        vpad = vb.double().view(1, 1, heads, 64).expand(nW, np_pad, heads, 64)
        vbf = torch.cat([vbf, vpad], 1)
    P = torch.softmax(S_, -1)
    ref = torch.einsum("whqk,wkhd->wqhd", P, vbf).reshape(M, C)
    assert float(S_[0].max() - S_[0, :, :, :32].max()) > 8 * 0.69, "the planted spikes must exceed the first chunk's maximum by more than the threshold"
    err = relerr(values(out, prec), ref)
    assert err < (2e-2 if prec == "bf16" else 1e-4), err
