"""CPU: the oracle restatement (oracle/toc3d_oracle.py) against the golden vectors produced by the REAL
reference (oracle/gen_golden.py, run in the build container).  This is the pin of the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import toc3d_oracle as O
from toc3d_amd import configs, synth


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _run_toc3d(cfg, sd, inp, prev, cap=None):
    with torch.no_grad():
        return O.forward_toc3d(sd, cfg, inp["x"], inp["temp_queries"], inp["temp_ref_points"], inp["temp_vel"], inp["temp_timestamp"],
                               inp["temp_ego_pose"], inp["ego_pose_inv"], prev, inp["gumbel"], cap)


@pytest.mark.parametrize("tag,prev,epoch", [("prev", True, False), ("first", False, False), ("prev_epoch", True, True)])
def test_tiny_toc3d_matches_reference(golden_dir, tag, prev, epoch):
    cfg = configs.get("toc3d_tiny")
    sd = synth.make_state_dict(cfg)
    inp = synth.make_inputs(cfg, views_per_frame=2, epoch_timestamps=epoch)
    g = _g(golden_dir, f"tiny_toc3d_{tag}")
    cap = {}
    out = _run_toc3d(cfg, sd, inp, prev, cap)
    ref = torch.from_numpy(g["last_feat"])
    assert (out["last_feat"] - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    for s in range(3):
        assert np.array_equal(out["keep_idx"][s].numpy(), g[f"keep_idx{s}"])
        assert np.array_equal(out["drop_idx"][s].numpy(), g[f"drop_idx{s}"])
        np.testing.assert_allclose(out["token_masks"][s].numpy(), g[f"token_mask{s}"], atol=1e-6)
    for k in g.files:
        if k.startswith("block"):
            np.testing.assert_allclose(cap[k].numpy(), g[k], rtol=0, atol=1e-4)


def test_tiny_eva_and_neck(golden_dir):
    cfg = configs.get("eva_tiny")
    sd = synth.make_state_dict(cfg)
    inp = synth.make_inputs(cfg, views_per_frame=2)
    with torch.no_grad():
        o = O.forward_eva(sd, cfg, inp["x"])
    ref = _g(golden_dir, "tiny_eva")["last_feat"]
    np.testing.assert_allclose(o["last_feat"].numpy(), ref, atol=1e-4)
    nsd = synth.neck_state_dict(configs.CPFPN_TINY)
    n0, n1 = O.cpfpn(nsd, o["last_feat"])
    gn = _g(golden_dir, "tiny_neck")
    np.testing.assert_allclose(n0.numpy(), gn["level0"], atol=1e-4)
    np.testing.assert_allclose(n1.numpy(), gn["level1"], atol=1e-4)


def test_units(golden_dir):
    g = _g(golden_dir, "units")
    t = lambda k: torch.from_numpy(g[k])
    for ws in (16, 20):
        w, pad = O.window_partition(t(f"wp{ws}.in"), ws)
        assert torch.equal(w, t(f"wp{ws}.out"))
        s, _ = O.window_partition(t(f"wp{ws}.in")[..., :1], ws, pad_value=-1e6)
        assert torch.equal(s, t(f"wp{ws}.score"))
        assert torch.equal(O.window_unpartition(w, ws, pad, (20, 50)), t(f"wp{ws}.unpart"))
    for hw in ((20, 50), (40, 100), (50, 100)):
        np.testing.assert_allclose(O.abs_pos(t("abs_pos.in"), True, hw).numpy(), g[f"abs_pos.{hw[0]}x{hw[1]}"], atol=1e-6)
    c16, s16 = synth.rope_tables(16)
    c20, s20 = synth.rope_tables(20)
    assert torch.equal(c16, t("rope16.cos")) and torch.equal(s16, t("rope16.sin"))
    assert torch.equal(c20, t("rope20.cos")) and torch.equal(s20, t("rope20.sin"))
    np.testing.assert_allclose(O.rope_rotate(t("rope16.t"), c16, s16).numpy(), g["rope16.applied"], atol=1e-6)
    sel = t("rope20.sel")
    np.testing.assert_allclose(O.rope_rotate(t("rope20.t"), c20[sel][:, None], s20[sel][:, None]).numpy(), g["rope20.applied"], atol=1e-6)
    idx = t("sel.idx")
    assert torch.equal(O.gather_rows(t("sel.x"), idx[:, :12]), t("sel.gather"))
    m = O.merge_tokens(O.gather_rows(t("sel.x"), idx[:, 12:]), O.gather_rows(t("merge.score"), idx[:, 12:]))
    np.testing.assert_allclose(m.numpy(), g["merge.out"], atol=1e-6)
    np.testing.assert_allclose(O.pos2posemb3d(t("pe3d.in")).numpy(), g["pe3d.out"], atol=1e-6)
    np.testing.assert_allclose(O.pos2posemb1d(t("pe1d.in")).numpy(), g["pe1d.out"], atol=1e-9)
    np.testing.assert_allclose(O.nerf_encoding(t("nerf.in")).numpy(), g["nerf.out"], atol=1e-6)


def test_scorer_stage(golden_dir):
    g = _g(golden_dir, "scorer_toc3d_tiny")
    cfg = configs.get("toc3d_tiny")
    sd = synth.make_state_dict(cfg)
    pre = "score_predictor.1."
    x = torch.from_numpy(synth._rng("scorer/x").standard_normal((2, 20, 50, cfg["embed_dim"]), dtype=np.float32))
    m = torch.from_numpy(synth._rng("scorer/m").random((2, 20, 50, 1), dtype=np.float32))
    for flavour, epoch in (("u01", False), ("epoch", True)):
        inp = synth.make_inputs(cfg, views_per_frame=2, epoch_timestamps=epoch)
        mq = O.motion_aware_queries(sd, pre, inp["temp_queries"], inp["temp_ref_points"], inp["temp_vel"], inp["temp_timestamp"],
                                    inp["temp_ego_pose"], inp["ego_pose_inv"])
        np.testing.assert_allclose(mq.numpy(), g[f"{flavour}.mq"], atol=1e-5)
        pq = O.query_based_score(x, m, mq, sd, pre)
        np.testing.assert_allclose(pq.numpy(), g[f"{flavour}.pred_query"], atol=1e-5)
        np.testing.assert_allclose(O.score_based_score(x, m, sd, pre).numpy(), g[f"{flavour}.pred_score"], atol=1e-5)
        ki, di, nm = O.sample_image_level(pq, cfg["token_ratio"][1], inp["gumbel"][1])
        assert np.array_equal(ki.numpy(), g[f"{flavour}.keep_idx"]) and np.array_equal(di.numpy(), g[f"{flavour}.drop_idx"])
        np.testing.assert_allclose(nm.numpy(), g[f"{flavour}.mask"], atol=1e-6)


def test_state_dict_spec_matches_reference(golden_dir):
    spec = json.load(open(os.path.join(golden_dir, "state_dict_spec.json")))
    for name in ("toc3d_tiny", "eva_tiny", "toc3d_faster", "eva_dense"):
        mine = {k: list(v) for k, v in synth.state_dict_spec(configs.get(name)).items()}
        assert mine == spec[name]
    n = sum(int(np.prod(v)) for k, v in spec["toc3d_faster"].items() if "freqs_" not in k and not k.endswith("pc_range"))
    assert abs(n - 311.06e6) < 0.05e6          # 311.06 M parameters


def test_image_oracle_normalise_pad_format():
    """oracle/image_oracle.py (parity unpinned: mmcv / OpenCV absent) against the published formula in float64, and the
    pipeline's layout rules: padded pixels are exactly zero, channel flip happens before mean/std are applied."""
    from oracle import image_oracle as I
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (2, 30, 50, 3), dtype=np.uint8)
    mean, std = [103.530, 116.280, 123.675], [57.375, 57.120, 58.395]          # projects/configs/ToC3D/ToC3D_faster.py:13-14
    for to_rgb in (False, True):
        out = I.prepare_images(img, mean, std, to_rgb, 32)
        assert out.shape == (2, 3, 32, 64) and out.dtype == np.float32
        src = img[..., ::-1] if to_rgb else img
        ref = (src.astype(np.float64) - np.float32(mean).astype(np.float64)) / np.float32(std).astype(np.float64)
        assert np.abs(out[:, :, :30, :50] - ref.transpose(0, 3, 1, 2)).max() < 4e-7 * 4.5       # <= 2 ulp of the largest value
        assert (out[:, :, 30:, :] == 0).all() and (out[:, :, :, 50:] == 0).all()
    # integer-valued float32 input (what LoadMultiViewImageFromFiles(to_float32=True) hands over) gives the same bits
    assert np.array_equal(I.imnormalize(img[0].astype(np.float32), mean, std, False), I.imnormalize(img[0], mean, std, False))


def test_image_oracle_against_the_reference_pipeline_classes(golden_dir):
    """oracle/image_oracle.py bit-exact against tests/golden/image_norm.npz: the reference's own NormalizeMultiviewImage.__call__ / PadMultiViewImage._pad_img
    (transform_3d.py:87-100,38-50) executed by oracle/gen_golden_image.py with the configs' mean / std / size_divisor, ragged sizes, both channel orders.
    The classes are pinned by this; mmcv.imnormalize / impad_to_multiple themselves were stand-ins written from the published definitions (mmcv / OpenCV are
    absent): that arithmetic stays unpinned upstream, as the fixture's `pinned` field says."""
    from oracle import image_oracle as I
    g = np.load(os.path.join(golden_dir, "image_norm.npz"))
    assert "stand-ins" in str(g["pinned"]) and "executed" in str(g["pinned"])
    mean, std, div = g["mean"].tolist(), g["std"].tolist(), int(g["size_divisor"])
    for tag in "abcd":
        u8, to_rgb, exp = g[f"{tag}_u8"], bool(g[f"{tag}_to_rgb"]), g[f"{tag}_expected"]
        out = I.prepare_images(u8, mean, std, to_rgb, div)
        assert out.shape == exp.shape and out.dtype == np.float32
        assert np.array_equal(out.view(np.uint32), exp.view(np.uint32)), f"case {tag}: the oracle differs from the reference pipeline's output"


def test_memory_bank_oracle_matches_reference_golden(golden_dir):
    """oracle/memory_oracle.py against tests/golden/memory_bank.npz, written by the reference's own
    StreamPETRHead.pre/post_update_memory (oracle/gen_golden_memory.py): bit-exact over a 4-frame sequence incl. a scene change."""
    from oracle.memory_oracle import MemoryBank
    from oracle.gen_golden_memory import CFG, B, NQ, NCLS, FRAMES
    g = np.load(os.path.join(golden_dir, "memory_bank.npz"))
    inp = synth.memory_inputs(CFG, B, NQ, NCLS, FRAMES, seed=0)
    m = MemoryBank(pseudo_reference_points=inp["pseudo"], **CFG)
    for f in range(FRAMES):
        fr = inp["frames"][f]
        m.pre_update_memory(fr["data"])
        for k, v in m.state().items():
            ref = torch.from_numpy(g[f"f{f}_pre_{k}"])
            assert v.dtype == ref.dtype and torch.equal(v, ref), (f, "pre", k)
        q = m.backbone_queries(8, mid_frame=f > 0)
        assert q["temp_queries"].shape[1] == 8 and (f > 0 or float(q["temp_queries"].abs().sum()) == 0.0)
        m.post_update_memory(fr["data"], fr["rec_ego_pose"], fr["cls"], fr["bbox"], fr["dec"])
        for k, v in m.state().items():
            ref = torch.from_numpy(g[f"f{f}_post_{k}"])
            assert v.dtype == ref.dtype and torch.equal(v, ref), (f, "post", k)


def test_head_tokens_oracle_matches_reference_golden(golden_dir):
    """oracle/head_tokens_oracle.py against tests/golden/head_tokens.npz, written by the reference's own position_embeding, MLN and
    SELayer_Linear composed as StreamPETRHead.forward :627-639 (oracle/gen_golden_head.py)."""
    from oracle import head_tokens_oracle as HO
    from oracle.gen_golden_head import CFG, B, N, H, W
    g = np.load(os.path.join(golden_dir, "head_tokens.npz"))
    sd = synth.head_tokens_state_dict(CFG)
    inp = synth.head_tokens_inputs(CFG, B, N, H, W)
    pad_h, pad_w = H * CFG["stride"], W * CFG["stride"]
    pos, cone = HO.position_embedding(sd, CFG, inp["intrinsics"], inp["lidar2img"], H, W, pad_h, pad_w)
    memory, pos_embed = HO.token_embeddings(sd, CFG, inp["feats"], inp["intrinsics"], inp["lidar2img"], pad_h, pad_w)
    for name, got in (("pos_raw", pos), ("cone", cone), ("memory", memory), ("pos_embed", pos_embed)):
        ref = torch.from_numpy(g[name])
        assert got.shape == ref.shape, name
        assert torch.equal(got, ref), (name, float((got - ref).abs().max()))
