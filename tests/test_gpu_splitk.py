"""GPU: deterministic split-K of the residual GEMMs (toc3d_linear_fused_ws, include/toc3d.h) -- attn.proj + residual (eva_vit.py:115,262 /
toc3d_eva_vit.py:514,379) and mlp.w3 + residual (eva_vit.py:49,263 / toc3d_eva_vit.py:384).

What is pinned: (1) a split launch equals the unsplit one up to the order of the K sum (f32 accumulators: <= a few ulp of the largest output),
every secondary output of the epilogues included (representative rows, act-dtype copy, statistics); (2) for one `split`, every tile variant returns
the SAME bits (the cuts of K are the same for every variant); (3) the bits do not depend on which slice arrives last: hundreds of reruns under
uneven load on a second stream, two shapes alternating on ONE workspace (the tickets re-arm themselves), every word compared.
"""
import pytest
import torch

from toc3d_amd import lib
from test_gpu_ops import DEV, S, as_act, pack, relerr, rnd

pytestmark = pytest.mark.gpu

SK_TILES = {lib.BF16: (1, 9, 10, 14, 16, 17, 19, 22, 26, 28, 29, 55, 56), lib.F32: (1, 9, 10, 14, 16, 17, 19, 22, 26, 28), lib.F32X3: (1, 9, 10, 14, 16, 17, 19, 22, 26, 28)}
DT = [("bf16", lib.BF16, torch.bfloat16), ("fp32", lib.F32, torch.float32), ("fp32x3", lib.F32X3, torch.float32)]


def workspace(nbytes):
    ws = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=DEV)          # tickets zero before the first launch
    assert ws.data_ptr() % 256 == 0
    return ws


def ws_bytes(variant, M, N):
    return lib.load().toc3d_linear_splitk_workspace_bytes(variant, M, N)


def _bits(t):
    return t.contiguous().view(torch.uint8)


@pytest.mark.parametrize("name,dt,tdt", DT)
@pytest.mark.parametrize("M,N,K", [(777, 640, 512), (2178, 1024, 2752), (300, 1024, 1024)])
def test_splitk_residual_epilogues_match_the_unsplit_launch(name, dt, tdt, M, N, K):
    pdt = lib.F32 if dt == lib.F32X3 else dt
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3).to(DEV)
    a_d, w_d = as_act(A, tdt), pack(W, pdt, tdt)
    x0 = (2.0 * rnd(M, N, seed=4) + 0.3).to(DEV)
    rep_index = torch.full((M,), -1, dtype=torch.int32, device=DEV)
    rep_index[::37] = torch.arange(len(range(0, M, 37)), dtype=torch.int32, device=DEV)
    nrep = int((rep_index >= 0).sum())
    # residual read through an index (compact rows whose residual still sits in another buffer), every 5th row in place
    src = (1.5 * rnd(M + 11, N, seed=5)).to(DEV)
    res_index = torch.arange(M, dtype=torch.int32, device=DEV) + 7
    res_index[::5] = -1
    cap2 = (N + 63) // 64
    folded = dt != lib.F32                                 # the folded-LayerNorm epilogues: bf16 and bf16 x 3

    def run(variant, epi, ws):
        x = x0.clone()
        rep = torch.zeros(nrep, N, device=DEV)
        extra = [None, 0, None, 0, None, 0, 0.0, None, 0, None]
        outs = [x, rep]
        if epi == lib.EPI_RESIDUAL:
            extra[9] = res_index
            args = (dt, epi, variant, a_d, K, w_d, K, b, x, N, src, N, 0, rep, rep_index, M, N, K, 0, *extra)
        elif epi == lib.EPI_RESIDUAL_STATS:
            a_raw = torch.full((M, N), 9.0, dtype=tdt, device=DEV)
            st = torch.zeros(4 + M * cap2 * 2, device=DEV)
            extra[0], extra[1], extra[7], extra[8] = st, cap2, a_raw, N
            outs += [a_raw, st]
            args = (dt, epi, variant, a_d, K, w_d, K, b, x, N, x, N, 0, rep, rep_index, M, N, K, 0, *extra)
        else:                                              # EPI_RESIDUAL_LN: statistics of the A rows as a producing GEMM leaves them (slots of 128 columns)
            slots = (K + 127) // 128
            af = a_d.float()
            st_in = torch.zeros(4 + M * slots * 2, device=DEV)
            st_in[:1].view(torch.int32)[0] = slots
            pad = torch.zeros(M, slots * 128, device=DEV)
            pad[:, :K] = af
            v = st_in[4:].view(M, slots, 2)
            v[..., 0] = pad.view(M, slots, 128).sum(2)
            v[..., 1] = (pad.view(M, slots, 128) ** 2).sum(2)
            c1 = rnd(N, seed=8).to(DEV)
            extra[2], extra[3], extra[4], extra[5], extra[6] = st_in, slots | slots << 32, c1, K - 3, 1e-6
            args = (dt, epi, variant, a_d, K, w_d, K, b, x, N, x, N, 0, rep, rep_index, M, N, K, 0, *extra)
        if variant >= 1000:
            lib.call("toc3d_linear_fused_ws", *args, ws, ws.numel() * 4, S())
        else:
            lib.call("toc3d_linear_fused", *args, S())
        return outs

    epis = [lib.EPI_RESIDUAL] + ([lib.EPI_RESIDUAL_STATS, lib.EPI_RESIDUAL_LN] if folded else [])
    for epi in epis:
        ref = run(16, epi, None)
        scale = ref[0].abs().max().item()
        for split in (2, 3, 4):
            if K < 128 * split:
                continue
            first = None
            for tv in SK_TILES[dt]:
                v = 1000 * split + tv
                need = ws_bytes(v, M, N)
                assert need > 65536
                ws = workspace(need)
                try:
                    got = run(v, epi, ws)
                except RuntimeError as e:                  # K is not a whole number of this variant's K-tiles (BK = 128 on K = 2752)
                    assert "K-tiles" in str(e) and K % 128 != 0 and tv in (22, 26), (v, str(e))
                    continue
                assert torch.count_nonzero(ws[:16384]) == 0, f"variant {v}: tickets not re-armed"
                # against the unsplit launch: the f32 stream within a few ulp of its largest value, act-dtype copy within one rounding
                assert (got[0] - ref[0]).abs().max().item() <= 4e-6 * scale, f"epi {epi} variant {v}"
                assert (got[1] - ref[1]).abs().max().item() <= 4e-6 * scale
                if epi == lib.EPI_RESIDUAL_STATS:
                    assert torch.equal(got[2], got[0].to(tdt)) if tdt == torch.bfloat16 else torch.equal(got[2], got[0])
                    # (the statistics sum ROUNDED values: an output within a few f32 ulp of a bf16 rounding boundary may round the other way)
                    assert relerr(got[3][4:], ref[3][4:]) < (2e-3 if tdt == torch.bfloat16 else 1e-5)
                if first is None:
                    first = got
                else:
                    for g, f in zip(got, first):
                        assert torch.equal(_bits(g), _bits(f)), f"epi {epi}: split {split} depends on the tile variant ({tv})"


def test_splitk_is_bit_stable_whoever_arrives_last():
    """400 reruns of two shapes alternating on ONE workspace, a GEMM stream of another shape beside them (uneven load: the arrival order of a
    tile's slices changes from run to run); every output word of every rerun equals the first run's."""
    dt, tdt = lib.BF16, torch.bfloat16
    C, Hp = 1024, 2752
    shapes = [(3744, C, C, 4016), (2178, C, Hp, 3017), (6000, C, Hp, 2029), (2898, C, C, 2056)]
    bufs = []
    need = 0
    for i, (M, N, K, v) in enumerate(shapes):
        a_d = as_act(rnd(M, K, seed=10 + i), tdt)
        w_d = pack(rnd(N, K, seed=20 + i, scale=K ** -0.5), dt, tdt)
        b = rnd(N, seed=30 + i).to(DEV)
        x0 = rnd(M, N, seed=40 + i).to(DEV)
        bufs.append((a_d, w_d, b, x0))
        need = max(need, ws_bytes(v, M, N))
    ws = workspace(need)
    # the co-runner
    Mo = 5000
    ao, wo, bo = as_act(rnd(Mo, C, seed=50), tdt), pack(rnd(3 * C, C, seed=51, scale=C ** -0.5), dt, tdt), rnd(3 * C, seed=52).to(DEV)
    oo = torch.zeros(Mo, 3 * C, dtype=tdt, device=DEV)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    cap2 = C // 64
    first = {}
    for it in range(100):
        if it % 3 != 2:
            with torch.cuda.stream(side):
                for _ in range(1 + it % 4):
                    lib.call("toc3d_linear_ex", dt, lib.EPI_BIAS, 16 if it % 2 else 17, ao, C, wo, C, bo, oo, 3 * C, None, 0, 0, None, None, Mo, 3 * C, C, 0, S())
        for i, (M, N, K, v) in enumerate(shapes):
            a_d, w_d, b, x0 = bufs[i]
            x = x0.clone()
            a_raw = torch.empty(M, N, dtype=tdt, device=DEV)
            st = torch.zeros(4 + M * cap2 * 2, device=DEV)
            lib.call("toc3d_linear_fused_ws", dt, lib.EPI_RESIDUAL_STATS, v, a_d, K, w_d, K, b, x, N, x, N, 0, None, None, M, N, K, 0,
                     st, cap2, None, 0, None, 0, 0.0, a_raw, N, None, ws, ws.numel() * 4, S())
            if it == 0:
                first[i] = (x, a_raw, st)
            else:
                for g, f, what in zip((x, a_raw, st), first[i], ("f32 stream", "act copy", "statistics")):
                    assert torch.equal(_bits(g), _bits(f)), f"rerun {it}, shape {shapes[i]}: {what} differs"
    torch.cuda.synchronize()
    assert torch.count_nonzero(ws[:16384]) == 0


def test_splitk_argument_checks():
    dt, tdt = lib.BF16, torch.bfloat16
    M, N, K = 300, 256, 512
    a_d, w_d, b = as_act(rnd(M, K, seed=1), tdt), pack(rnd(N, K, seed=2), dt, tdt), rnd(N, seed=3).to(DEV)
    x = torch.zeros(M, N, device=DEV)
    args = (a_d, K, w_d, K, b, x, N, x, N, 0, None, None, M, N, K, 0, *lib.NO_FUSED)
    ws = workspace(ws_bytes(4016, M, N))
    assert ws_bytes(16, M, N) < 0 and ws_bytes(5016, M, N) < 0 and ws_bytes(2008, M, N) < 0
    with pytest.raises(RuntimeError, match="workspace"):
        lib.call("toc3d_linear_fused_ws", dt, lib.EPI_RESIDUAL, 2016, *args, ws, 1024, S())
    with pytest.raises(RuntimeError, match="workspace"):
        lib.call("toc3d_linear_fused_ws", dt, lib.EPI_RESIDUAL, 2016, *args, None, 0, S())
    with pytest.raises(RuntimeError, match="no split-K form"):
        lib.call("toc3d_linear_fused_ws", dt, lib.EPI_RESIDUAL, 2008, *args, ws, ws.numel() * 4, S())
    with pytest.raises(RuntimeError, match="residual epilogues"):
        lib.call("toc3d_linear_fused_ws", dt, lib.EPI_BIAS, 2016, *args, ws, ws.numel() * 4, S())
    with pytest.raises(RuntimeError, match="takes a workspace"):
        lib.call("toc3d_linear_fused", dt, lib.EPI_RESIDUAL, 2016, *args, S())
    with pytest.raises(RuntimeError, match="too short"):
        lib.call("toc3d_linear_fused_ws", dt, lib.EPI_RESIDUAL, 4016, a_d, K, w_d, K, b, x, N, x, N, 0, None, None, M, N, 256, 0, *lib.NO_FUSED, ws, ws.numel() * 4, S())
    # an unsplit variant through the workspace entry point is the plain launch
    y = torch.zeros(M, N, device=DEV)
    lib.call("toc3d_linear_fused_ws", dt, lib.EPI_RESIDUAL, 16, a_d, K, w_d, K, b, y, N, None, 0, 0, None, None, M, N, K, 0, *lib.NO_FUSED, None, 0, S())
    z = torch.zeros(M, N, device=DEV)
    lib.call("toc3d_linear_fused", dt, lib.EPI_RESIDUAL, 16, a_d, K, w_d, K, b, z, N, None, 0, 0, None, None, M, N, K, 0, *lib.NO_FUSED, S())
    assert torch.equal(y, z)
