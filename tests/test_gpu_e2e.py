"""GPU: whole-backbone parity through the drop-in modules (which call only the C ABI) against golden vectors
produced by the REAL reference (tests/golden, see oracle/gen_golden.py) and against the oracle.

Parity bar: fp32 path <= 1e-3 relative (max-abs error / max-abs reference) with identical
kept-index sets; bf16 path is reported (relative L2, index IoU) and bounded loosely -- 1e-3 is not reachable
with bf16 operands by any implementation.
"""
import os

import numpy as np
import pytest
import torch

import toc3d_amd
from oracle import toc3d_oracle as O
from toc3d_amd import configs, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_max(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()


def iou(a, b):
    res = []
    for ra, rb in zip(a.cpu().numpy(), np.asarray(b)):
        sa, sb = set(ra.tolist()), set(rb.tolist())
        res.append(len(sa & sb) / max(1, len(sa | sb)))
    return min(res)


def build(name, precision):
    cfg = configs.get(name)
    m = toc3d_amd.build_backbone(dict(cfg, precision=precision))
    m.load_state_dict(synth.make_state_dict(cfg), strict=True)
    return cfg, m.to(DEV).eval()


def run_toc3d(m, inp, prev):
    d = lambda t: t.to(DEV)
    return m(d(inp["x"]), temp_queries=d(inp["temp_queries"]), prev_exists=prev, temp_ref_points=d(inp["temp_ref_points"]),
             temp_vel=d(inp["temp_vel"]), temp_timestamp=d(inp["temp_timestamp"]), temp_ego_pose=d(inp["temp_ego_pose"]),
             ego_pose_inv=d(inp["ego_pose_inv"]), gumbel_noise=inp["gumbel"],
             gt_bboxes=None, gt_centers2d=None, gt_depths=None)       # ignored extras the caller passes (petr3d.py:145-157)


@pytest.mark.parametrize("tag,prev,epoch", [("prev", True, False), ("first", False, False), ("prev_epoch", True, True)])
def test_tiny_toc3d_fp32_matches_reference(golden_dir, tag, prev, epoch):
    cfg, m = build("toc3d_tiny", "fp32")
    inp = synth.make_inputs(cfg, views_per_frame=2, epoch_timestamps=epoch)
    g = np.load(os.path.join(golden_dir, f"tiny_toc3d_{tag}.npz"))
    out = run_toc3d(m, inp, prev)
    assert isinstance(out, toc3d_amd.ToC3DViTReturnType) and out.attn_scores is None and out.aux_outputs is None
    feat = out.img_feats["last_feat"]
    assert tuple(feat.shape) == (2, 128, 20, 50) and not feat.is_contiguous()          # NCHW view of NHWC (toc3d_eva_vit.py:294)
    for s in range(3):
        assert iou(out.keep_idx[s], g[f"keep_idx{s}"]) > 0.995
        assert out.keep_idx[s].dtype == torch.int64 and out.keep_idx[s].shape == g[f"keep_idx{s}"].shape
        assert out.drop_idx[s].shape == g[f"drop_idx{s}"].shape
        assert (out.token_masks[s].cpu() - torch.from_numpy(g[f"token_mask{s}"])).abs().max().item() < 2e-3
    err = rel_max(feat, torch.from_numpy(g["last_feat"]))
    print(f"[tiny fp32 {tag}] rel max err {err:.3e}  rel l2 {rel_l2(feat, torch.from_numpy(g['last_feat'])):.3e}")
    assert err < 1e-3


def test_tiny_toc3d_fp32_indices_exact_and_rerun_deterministic(golden_dir):
    cfg, m = build("toc3d_tiny", "fp32")
    inp = synth.make_inputs(cfg, views_per_frame=2)
    g = np.load(os.path.join(golden_dir, "tiny_toc3d_prev.npz"))
    o1 = run_toc3d(m, inp, True)
    f1 = o1.img_feats["last_feat"].clone()
    o2 = run_toc3d(m, inp, True)
    assert torch.equal(f1, o2.img_feats["last_feat"]), "same inputs + same injected noise must be bit-identical"
    assert np.array_equal(o1.keep_idx[0].cpu().numpy(), g["keep_idx0"]) and np.array_equal(o1.drop_idx[0].cpu().numpy(), g["drop_idx0"])


@pytest.mark.parametrize("tag,prev", [("prev", True), ("first", False)])
def test_tiny_toc3d_bf16_reported(golden_dir, tag, prev):
    cfg, m = build("toc3d_tiny", "bf16")
    inp = synth.make_inputs(cfg, views_per_frame=2)
    g = np.load(os.path.join(golden_dir, f"tiny_toc3d_{tag}.npz"))
    out = run_toc3d(m, inp, prev)
    feat = out.img_feats["last_feat"]
    ious = [iou(out.keep_idx[s], g[f"keep_idx{s}"]) for s in range(3)]
    l2 = rel_l2(feat, torch.from_numpy(g["last_feat"]))
    print(f"[tiny bf16 {tag}] free-running rel l2 {l2:.3e} keep IoU {ious}")
    assert ious[0] > 0.97 and l2 < 0.15 and torch.isfinite(feat).all()


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 5e-2)])
def test_tiny_eva_and_neck(golden_dir, precision, tol):
    cfg, m = build("eva_tiny", precision)
    inp = synth.make_inputs(cfg, views_per_frame=2)
    out = m(inp["x"].to(DEV))
    ref = torch.from_numpy(np.load(os.path.join(golden_dir, "tiny_eva.npz"))["last_feat"])
    err = rel_max(out["last_feat"], ref)
    print(f"[tiny eva {precision}] rel max err {err:.3e}")
    assert err < tol
    neck = toc3d_amd.build_neck(dict(configs.CPFPN_TINY, precision=precision))
    neck.load_state_dict(synth.neck_state_dict(configs.CPFPN_TINY))
    neck = neck.to(DEV)
    gn = np.load(os.path.join(golden_dir, "tiny_neck.npz"))
    n0, n1 = neck([ref.to(DEV)])                                       # feed the reference features: isolates the neck
    assert n0.is_contiguous() and tuple(n0.shape) == (2, 32, 20, 50) and tuple(n1.shape) == (2, 32, 10, 25)
    e0 = rel_max(n0, torch.from_numpy(gn["level0"]))
    print(f"[tiny neck {precision}] rel max err {e0:.3e}")
    assert e0 < (1e-4 if precision == "fp32" else 3e-2)
    assert rel_max(n1, torch.from_numpy(gn["level1"])) < (1e-4 if precision == "fp32" else 3e-2)
    n0b, _ = neck([out["last_feat"]])                                  # zero-copy path from the backbone's NHWC buffer
    assert rel_max(n0b, torch.from_numpy(gn["level0"])) < (2e-3 if precision == "fp32" else 1e-1)


@pytest.mark.parametrize("precision", ["fp32", "fp32x3", "fp32x6"])
@pytest.mark.parametrize("name", ["toc3d_faster", "toc3d_fast", "eva_dense"])
def test_vitl_fp32_matches_reference(golden_dir, name, precision):
    """Full-size EVA-02 ViT-L configs of BASELINE.json (6 views @ 800x320) on the strict-parity path (exact-f32 MFMA) and on the
    parity-grade fast path (f32 buffers, products as three bf16 MFMAs): both inside the 1e-3 bar against the REAL reference's golden."""
    cfg, m = build(name, precision)
    inp = synth.make_inputs(cfg, views_per_frame=6)
    g = np.load(os.path.join(golden_dir, f"vitl_{name}.npz"))
    if synth.is_toc3d(cfg):
        out = run_toc3d(m, inp, True)
        feat = out.img_feats["last_feat"]
        for s in range(3):
            print(f"[vitl {name}] stage {s} keep IoU {iou(out.keep_idx[s], g[f'keep_idx{s}']):.4f}")
            assert iou(out.keep_idx[s], g[f"keep_idx{s}"]) > 0.99
            assert (out.token_masks[s][..., 0].cpu() - torch.from_numpy(g[f"token_mask{s}"])).abs().max().item() < 5e-3
    else:
        feat = m(inp["x"].to(DEV))["last_feat"]
    ref = torch.from_numpy(g["last_feat.c16"])
    err, l2 = rel_max(feat[:, ::16], ref), rel_l2(feat[:, ::16], ref)
    tl2 = (feat.double().norm(dim=1).cpu() - torch.from_numpy(g["last_feat.token_l2"])).abs().max().item() / g["last_feat.token_l2"].max()
    print(f"[vitl {name} {precision}] rel max err {err:.3e} rel l2 {l2:.3e} token-norm err {tl2:.3e}")
    assert err < 1e-3 and tl2 < 1e-3


@pytest.mark.parametrize("name", ["toc3d_faster", "eva_dense"])
def test_vitl_bf16_reported(golden_dir, name):
    cfg, m = build(name, "bf16")
    inp = synth.make_inputs(cfg, views_per_frame=6)
    g = np.load(os.path.join(golden_dir, f"vitl_{name}.npz"))
    if synth.is_toc3d(cfg):
        out = run_toc3d(m, inp, True)
        feat = out.img_feats["last_feat"]
        print(f"[vitl {name} bf16] keep IoU per stage {[round(iou(out.keep_idx[s], g[f'keep_idx{s}']), 4) for s in range(3)]}")
    else:
        feat = m(inp["x"].to(DEV))["last_feat"]
    ref = torch.from_numpy(g["last_feat.c16"])
    l2 = rel_l2(feat[:, ::16], ref)
    print(f"[vitl {name} bf16] free-running rel l2 vs fp32 reference {l2:.3e}")
    # dense: no discrete decisions, pure bf16 arithmetic (measured 2.1e-2).  ToC3D free-running: 0.118 measured, top-k flips included;
    # the arithmetic part alone (forced selection) and the torch-bf16 control are bounded in tests/test_gpu_parity_bf16.py
    assert torch.isfinite(feat).all() and l2 < (3e-2 if name == "eva_dense" else 0.16)


def test_state_dict_reload_repacks(golden_dir):
    cfg, m = build("eva_tiny", "fp32")
    inp = synth.make_inputs(cfg, views_per_frame=2)
    a = m(inp["x"].to(DEV))["last_feat"].clone()
    sd2 = synth.make_state_dict(cfg, seed=1)
    m.load_state_dict(sd2)
    b = m(inp["x"].to(DEV))["last_feat"].clone()
    with torch.no_grad():
        ref = O.forward_eva(sd2, cfg, inp["x"])["last_feat"]
    assert not torch.allclose(a, b) and rel_max(b, ref) < 1e-3


@pytest.mark.parametrize("name", ["toc3d_tiny", "eva_tiny"])
def test_view_groups_on_separate_streams_are_bit_identical(name):
    """Views are independent units: splitting them into concurrently running groups must not change a single bit."""
    cfg, m = build(name, "bf16")
    inp = synth.make_inputs(cfg, views_per_frame=2)
    run = (lambda: run_toc3d(m, inp, True).img_feats["last_feat"].clone()) if synth.is_toc3d(cfg) else (lambda: m(inp["x"].to(DEV))["last_feat"].clone())
    a = run()
    m.view_groups = 2
    b = run()
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    if synth.is_toc3d(cfg):
        o1 = run_toc3d(m, inp, True)
        m.view_groups = 1
        o2 = run_toc3d(m, inp, True)
        for s in range(3):
            assert torch.equal(o1.keep_idx[s], o2.keep_idx[s]) and torch.equal(o1.token_masks[s], o2.token_masks[s])


@pytest.mark.parametrize("name,groups,instances,vpf", [("toc3d_tiny", 2, 10, 2), ("toc3d_tiny", 2, 4, 6), ("toc3d_faster", 1, 1, 6),
                                                       ("toc3d_faster", 2, 1, 6)])
def test_repeated_forwards_are_bit_identical(name, groups, instances, vpf):
    """Same weights, same inputs, same injected noise: every forward returns the same bits -- across freshly built
    models (new stream objects / allocations) and with view groups overlapping on separate streams.  Regression for the
    packed-FP32 erratum (LABNOTES.md) that made ~40% of the two-group bf16 runs differ in the last bits."""
    ref = None
    for _ in range(instances):
        cfg, m = build(name, "bf16")
        m.view_groups, m.autotune = groups, False
        inp = synth.make_inputs(cfg, views_per_frame=vpf)
        for _ in range(4):
            out = run_toc3d(m, inp, True)
            cur = (out.img_feats["last_feat"].clone(), [k.clone() for k in out.keep_idx], [t.clone() for t in out.token_masks])
            torch.cuda.synchronize()
            if ref is None:
                ref = cur
            assert torch.equal(cur[0], ref[0])
            for s in range(3):
                assert torch.equal(cur[1][s], ref[1][s]) and torch.equal(cur[2][s], ref[2][s])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("hw", [(320, 800), (300, 790)])
def test_uint8_images_give_the_same_bits_as_host_normalised_float(precision, hw):
    """raw uint8 HWC camera images + img_norm_cfg == the reference's host pipeline
    (NormalizeMultiviewImage -> PadMultiViewImage -> CHW float32, restated in oracle/image_oracle.py) fed as float32."""
    from oracle import image_oracle as I
    norm = dict(mean=[103.530, 116.280, 123.675], std=[57.375, 57.120, 58.395], to_rgb=False)
    cfg = configs.get("toc3d_tiny")
    m = toc3d_amd.build_backbone(dict(cfg, precision=precision, img_norm_cfg=norm))
    m.load_state_dict(synth.make_state_dict(cfg), strict=True)
    m = m.to(DEV).eval()
    inp = synth.make_inputs(cfg, views_per_frame=2)
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (2, hw[0], hw[1], 3), dtype=np.uint8)
    xf = torch.from_numpy(I.prepare_images(img, norm["mean"], norm["std"], norm["to_rgb"], 32))
    a = run_toc3d(m, dict(inp, x=xf), True)
    fa, ka = a.img_feats["last_feat"].clone(), [k.clone() for k in a.keep_idx]
    b = run_toc3d(m, dict(inp, x=torch.from_numpy(img)), True)
    assert fa.shape == b.img_feats["last_feat"].shape == (2, cfg["embed_dim"], 320 // 16, 800 // 16)
    assert torch.equal(fa, b.img_feats["last_feat"])
    for s in range(3):
        assert torch.equal(ka[s], b.keep_idx[s])
    plain = toc3d_amd.build_backbone(dict(cfg, precision=precision)).to(DEV).eval()
    with pytest.raises(ValueError, match="img_norm_cfg"):
        plain(torch.from_numpy(img).to(DEV))


@pytest.mark.parametrize("name", ["toc3d_tiny", "eva_tiny"])
def test_packed_weight_cache_and_checkpoint_converter(name, tmp_path, golden_dir):
    """a reference-style .pth (keys img_backbone.*, tools/test.py:207) -> packed file -> a model that never sees the state dict.
    Checked against the REFERENCE (the golden features oracle/gen_golden.py wrote from the real backbone on the same synthetic checkpoint and inputs:
    fp32 <= 1e-3 with the reference's kept sets, bf16 inside its band), and against the model that was packed from the state dict directly
    (bit-identical, also at a resolution the writer never ran: abs-pos re-derived)."""
    from toc3d_amd.packed_io import convert_checkpoint
    cfg = configs.get(name)
    toc = synth.is_toc3d(cfg)
    sd = synth.make_state_dict(cfg)
    ckpt = {"meta": {}, "state_dict": {"img_backbone." + k: v for k, v in sd.items()} | {"pts_bbox_head.x": torch.zeros(1)}}
    torch.save(ckpt, tmp_path / "det.pth")
    g = np.load(os.path.join(golden_dir, "tiny_toc3d_prev.npz" if toc else "tiny_eva.npz"))
    gold = torch.from_numpy(g["last_feat"])
    inp = synth.make_inputs(cfg, views_per_frame=2)                 # the inputs the golden file was written from (oracle/gen_golden.py)
    inp2 = synth.make_inputs(cfg, views_per_frame=1, hw=(256, 512))
    fwd = (lambda m, i: run_toc3d(m, i, True)) if toc else (lambda m, i: m(i["x"].to(DEV)))
    feat = lambda o: (o.img_feats if toc else o)["last_feat"].clone()
    for precision in ("fp32", "bf16"):
        packed = str(tmp_path / f"det.backbone.{precision}.safetensors")
        a = convert_checkpoint(str(tmp_path / "det.pth"), dict(cfg, precision=precision), packed)
        ref = feat(fwd(a, inp))
        torch.manual_seed(123)
        b = toc3d_amd.build_backbone(dict(cfg, precision=precision)).to(DEV).eval()        # random weights, never loaded
        b.load_packed(packed)
        ob = fwd(b, inp)
        fb = feat(ob)
        assert torch.equal(fb, ref)
        assert torch.equal(feat(fwd(b, inp2)), feat(fwd(a, inp2)))
        # ... and the restored model reproduces the reference, not just its own writer
        if precision == "fp32":
            err = rel_max(fb, gold)
            print(f"[converter {name} fp32] restored-from-packed model vs the reference's features: rel max err {err:.3e}")
            assert err < 1e-3
            if toc:
                for st in range(3):
                    assert iou(ob.keep_idx[st], g[f"keep_idx{st}"]) > 0.995
        else:
            err = rel_l2(fb, gold)
            print(f"[converter {name} bf16] restored-from-packed model vs the reference's features: rel l2 {err:.3e}")
            assert err < (1.5e-1 if toc else 3e-2)                  # the free-running bf16 band of __graft_entry__.smoke() / the dense bf16 band
    c = toc3d_amd.build_backbone(dict(cfg, precision="fp32")).to(DEV).eval()
    with pytest.raises(ValueError, match="different model"):
        c.load_packed(packed)                                       # (the bf16 file)
    with pytest.raises(KeyError):
        convert_checkpoint(str(tmp_path / "det.pth"), dict(cfg, precision="bf16"), packed, prefix="backbone.")


def test_vitl_hires_1600x640_fp32_matches_oracle():
    """BASELINE.json config 4 geometry (40x100 tokens: 21 + 10 windows per view), 1 view, against the oracle run on the host."""
    cfg, m = build("toc3d_faster", "fp32")
    sd = synth.make_state_dict(cfg)
    inp = synth.make_inputs(cfg, views_per_frame=1, hw=(640, 1600))
    with torch.no_grad():
        ref = O.forward_toc3d(sd, cfg, inp["x"], inp["temp_queries"], inp["temp_ref_points"], inp["temp_vel"], inp["temp_timestamp"],
                              inp["temp_ego_pose"], inp["ego_pose_inv"], True, inp["gumbel"])
    out = run_toc3d(m, inp, True)
    feat = out.img_feats["last_feat"]
    assert tuple(feat.shape) == (1, 1024, 40, 100)
    for s in range(3):
        assert iou(out.keep_idx[s], ref["keep_idx"][s].numpy()) > 0.99
    err = rel_max(feat, ref["last_feat"])
    print(f"[vitl toc3d_faster 1600x640 fp32, 1 view] rel max err {err:.3e}")
    assert err < 1e-3


@pytest.mark.parametrize("groups,frames", [(1, 2), (2, 2), (1, 4)])
def test_two_frames_batch_matches_oracle(groups, frames):
    """B = 2 / 4 frames per forward (the scorer's queries are per frame, toc3d_utils.py:240 repeat_interleave; what bench.py --frames-total runs per rank):
    fp32 path vs the oracle; B = 4 stacked from four single-frame inputs the way bench.py does it (synth.stack_frames)."""
    cfg, m = build("toc3d_tiny", "fp32")
    m.view_groups = groups
    sd = synth.make_state_dict(cfg)
    if frames == 2:
        inp = synth.make_inputs(cfg, n_frames=2, views_per_frame=2)
    else:
        inp = synth.stack_frames([synth.make_inputs(cfg, n_frames=1, views_per_frame=2, seed=f) for f in range(frames)])
        assert inp["x"].shape[0] == 2 * frames and inp["temp_queries"].shape[0] == frames and inp["gumbel"][0].shape[0] == 2 * frames
    with torch.no_grad():
        ref = O.forward_toc3d(sd, cfg, inp["x"], inp["temp_queries"], inp["temp_ref_points"], inp["temp_vel"], inp["temp_timestamp"],
                              inp["temp_ego_pose"], inp["ego_pose_inv"], True, inp["gumbel"])
    out = run_toc3d(m, inp, True)
    assert rel_max(out.img_feats["last_feat"], ref["last_feat"]) < 1e-3
    for s in range(3):          # order among near-equal scores may differ by float rounding; the kept *sets* must agree
        assert iou(out.keep_idx[s], ref["keep_idx"][s].numpy()) > 0.995


# ---------------------------------------------------------------------------------------------------
# temporal memory bank
def _run_memory(mem, oracle, inp, frames, tol=2e-6):
    d = lambda t: t.to(DEV)
    for f in range(frames):
        fr = inp["frames"][f]
        data = {k: d(v) for k, v in fr["data"].items()}
        mem.pre_update_memory(data)
        oracle.pre_update_memory(fr["data"])
        for phase in ("pre", "post"):
            if phase == "post":
                mem.post_update_memory(data, d(fr["rec_ego_pose"]), d(fr["cls"])[None], d(fr["bbox"])[None], d(fr["dec"])[None])
                oracle.post_update_memory(fr["data"], fr["rec_ego_pose"], fr["cls"], fr["bbox"], fr["dec"])
            ref = oracle.state()
            got = dict(embedding=mem.memory_embedding, reference_point=mem.memory_reference_point, timestamp=mem.memory_timestamp,
                       egopose=mem.memory_egopose, velo=mem.memory_velo)
            for k in ref:
                r, g = ref[k].double(), got[k].double().cpu()
                assert r.shape == g.shape, (f, phase, k, r.shape, g.shape)
                if k in ("embedding", "velo", "timestamp"):              # copies, masks and f64 sums: exact
                    assert torch.equal(g, r), (f, phase, k)
                else:                                                    # 4x4 products: torch's CPU bmm may fuse multiply-adds
                    assert (g - r).abs().max() <= tol * max(1.0, float(r.abs().max())), (f, phase, k, float((g - r).abs().max()))


def test_memory_bank_matches_reference_golden_sequence(golden_dir):
    """toc3d_amd.TemporalMemory over the 4-frame golden sequence written by the reference's own head methods."""
    from oracle.memory_oracle import MemoryBank
    from oracle.gen_golden_memory import CFG, B, NQ, NCLS, FRAMES
    inp = synth.memory_inputs(CFG, B, NQ, NCLS, FRAMES, seed=0)
    mem = toc3d_amd.TemporalMemory(pseudo_reference_points=inp["pseudo"], **CFG)
    g = np.load(os.path.join(golden_dir, "memory_bank.npz"))
    d = lambda t: t.to(DEV)
    for f in range(FRAMES):
        fr = inp["frames"][f]
        data = {k: d(v) for k, v in fr["data"].items()}
        mem.pre_update_memory(data)
        mem.post_update_memory(data, d(fr["rec_ego_pose"]), d(fr["cls"])[None], d(fr["bbox"])[None], d(fr["dec"])[None])
        for k, got in (("embedding", mem.memory_embedding), ("velo", mem.memory_velo), ("timestamp", mem.memory_timestamp),
                       ("reference_point", mem.memory_reference_point), ("egopose", mem.memory_egopose)):
            ref = torch.from_numpy(g[f"f{f}_post_{k}"]).double()
            got = got.double().cpu()
            if k in ("embedding", "velo", "timestamp"):
                assert torch.equal(got, ref), (f, k)
            else:
                assert (got - ref).abs().max() <= 2e-6 * max(1.0, float(ref.abs().max())), (f, k)
    _run_memory(toc3d_amd.TemporalMemory(pseudo_reference_points=inp["pseudo"], **CFG), MemoryBank(pseudo_reference_points=inp["pseudo"], **CFG), inp, FRAMES)


def test_memory_bank_full_size_feeds_the_backbone():
    """Shipped sizes (memory_len 512, top-k 128, 128 propagated, 256-d, 900 + 128 queries, projects/configs/ToC3D/ToC3D_faster.py) against the
    oracle for 3 frames, then the bank's first 64 slots drive a ToC3D forward exactly like tensors sliced from the oracle's bank."""
    from oracle.memory_oracle import MemoryBank
    cfgm = dict(memory_len=512, topk_proposals=128, num_propagated=128, embed_dims=256, pc_range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0])
    inp = synth.memory_inputs(cfgm, 1, 900, 10, 3, seed=1)
    mem = toc3d_amd.TemporalMemory(pseudo_reference_points=inp["pseudo"], **cfgm)
    ora = MemoryBank(pseudo_reference_points=inp["pseudo"], **cfgm)
    _run_memory(mem, ora, inp, 3)
    with pytest.raises(RuntimeError, match="CUDA/HIP"):
        mem.pre_update_memory(inp["frames"][0]["data"])
    cfg, m = build("toc3d_tiny", "fp32")
    binp = synth.make_inputs(cfg, views_per_frame=2)
    q = mem.backbone_queries(cfg["pruning_num_queries"], prev_exists=True)
    oq = ora.backbone_queries(cfg["pruning_num_queries"], True)
    x, inv = binp["x"].to(DEV), binp["ego_pose_inv"].to(DEV)
    a = m(x, ego_pose_inv=inv, gumbel_noise=binp["gumbel"], **q)
    fa, ka = a.img_feats["last_feat"].clone(), [k.clone() for k in a.keep_idx]
    b = m(x, ego_pose_inv=inv, gumbel_noise=binp["gumbel"], prev_exists=True, **{k: v.to(DEV) for k, v in oq.items()})
    assert float((fa - b.img_feats["last_feat"]).abs().max()) < 1e-3 * float(fa.abs().max())
    for s in range(3):
        assert iou(ka[s], b.keep_idx[s].cpu()) > 0.99


def test_carried_compact_set_matches_scatter_gather_between_blocks():
    """Consecutive accelerated blocks of one window type continue on the same compact rows (no scatter + gather in between;
    bf16 path by default).  Proven on the fp32 kernels, where nothing else differs: features equal the block-by-block form to
    1e-5 and the token selections are identical; in bf16 the two forms differ like any two bf16 roundings (top-k flips)."""
    inp = synth.make_inputs(configs.get("toc3d_faster"), views_per_frame=6)
    for precision, tol, min_iou in (("fp32", 1e-5, 1.0), ("bf16", 1e-1, 0.95)):
        outs = {}
        for carry in (True, False):
            _, m = build("toc3d_faster", precision)
            m.carry_compact, m.autotune = carry, False
            o = run_toc3d(m, inp, True)
            outs[carry] = (o.img_feats["last_feat"].float().clone(), [k.clone() for k in o.keep_idx])
            del m
        a, b = outs[True], outs[False]
        rel = float((a[0] - b[0]).norm() / b[0].norm())
        print(f"[carry {precision}] rel l2 between carried and block-by-block features: {rel:.3e}")
        assert rel < tol, (precision, rel)
        for s in range(3):
            assert iou(a[1][s], b[1][s].cpu()) >= min_iou
    _, m32 = build("toc3d_tiny", "fp32")
    assert not m32.carry_compact, "the strict-parity path keeps the reference's scatter / gather between all blocks by default"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_split_merge_shares_one_scratch_between_window_types(precision):
    """gather_split=True end to end where the two window types of a plan have window counts in different 256-byte buckets of the old scratch layout
    (two frames per forward at 800x320: 96 local and 36 global windows; ADVICE r04: the small launch's partials overwrote counter words 64.. of the
    large one, whose representative rows were then never written).  The split merge must return the bits of the single-workgroup merge, on every
    frame of a replayed plan."""
    cfg = configs.get("toc3d_faster")
    inp = synth.stack_frames([synth.make_inputs(cfg, n_frames=1, views_per_frame=6, seed=f) for f in range(2)])
    outs = {}
    for split in (False, True, 16):
        _, m = build("toc3d_faster", precision)
        m.gather_split, m.autotune = split, False
        feats = []
        for _ in range(3):                                   # eager warm-up, recording, replay
            o = run_toc3d(m, inp, True)
            feats.append(o.img_feats["last_feat"].float().clone())
        assert torch.equal(feats[1], feats[2])
        outs[split] = (feats[2], [k.clone() for k in o.keep_idx])
        del m
    for split in (True, 16):
        assert torch.equal(outs[split][0], outs[False][0]), f"gather_split={split}: features differ from the single-workgroup merge"
        for s in range(3):
            assert torch.equal(outs[split][1][s], outs[False][1][s])


def test_carried_compact_set_on_long_runs_of_one_window_type():
    """A layout the shipped configs do not have: up to six consecutive accelerated blocks of one window type within a stage
    (global_attn_indexes=(2, 11), pruning_loc=[3, 9]).  Carried sets must pair up (3,4) (5,6) (7,8); a block after a pair starts from
    the scattered stream again -- regression for a pairing rule that lost the third block's update in runs of four or more."""
    cfg = dict(configs.get("toc3d_tiny"), global_attn_indexes=(2, 11), pruning_loc=[3, 9], token_ratio=[0.5, 0.3])
    sd = synth.make_state_dict(cfg)
    inp = synth.make_inputs(cfg, views_per_frame=2)
    inp["gumbel"] = inp["gumbel"][:2]
    with torch.no_grad():
        ref = O.forward_toc3d(sd, cfg, inp["x"], inp["temp_queries"], inp["temp_ref_points"], inp["temp_vel"], inp["temp_timestamp"],
                              inp["temp_ego_pose"], inp["ego_pose_inv"], True, inp["gumbel"])["last_feat"]
    outs = {}
    for carry in (True, False):
        m = toc3d_amd.build_backbone(dict(cfg, precision="fp32"))
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV).eval()
        m.carry_compact, m.autotune = carry, False
        pairs = []
        from toc3d_amd.testing import instrument
        instrument(m).block_hook = lambda i, gp, carried: pairs.append((i, carried))
        outs[carry] = run_toc3d(m, inp, True).img_feats["last_feat"].clone()
        if carry:
            assert [i for i, c in pairs if c] == [3, 5, 7, 9], pairs          # 9->10 pairs too; block 11 is global
    assert rel_max(outs[False], ref) < 1e-3
    assert rel_max(outs[True], ref) < 1e-3 and rel_max(outs[True], outs[False]) < 1e-4


def test_streaming_loop_uint8_images_memory_bank_backbone_neck():
    """All SURVEY.md 8f rows in the order the detector runs them (detectors/petr3d.py:105-190, streampetr_head.py:322-377):
    uint8 camera images -> memory bank slice -> ToC3D backbone -> CPFPN, then the bank absorbs (synthetic) head outputs; four frames
    with a scene start.  The fp32 HIP path tracks the oracle composition frame by frame; a second run of the loop returns the same bits."""
    from oracle import image_oracle as I
    from oracle.memory_oracle import MemoryBank
    cfg = configs.get("toc3d_tiny")
    norm = dict(mean=[103.530, 116.280, 123.675], std=[57.375, 57.120, 58.395], to_rgb=False)
    sd, nsd = synth.make_state_dict(cfg), synth.neck_state_dict(configs.CPFPN_TINY)
    mcfg = dict(memory_len=96, topk_proposals=32, num_propagated=32, embed_dims=256, pc_range=cfg["pc_range"])
    Q = cfg["pruning_num_queries"]
    minp = synth.memory_inputs(mcfg, 1, 120, 10, 4, seed=3)
    rng = np.random.default_rng(21)
    imgs = [rng.integers(0, 256, (2, 320, 800, 3), dtype=np.uint8) for _ in range(4)]
    base = synth.make_inputs(cfg, views_per_frame=2)

    def run_loop():
        bb = toc3d_amd.build_backbone(dict(cfg, precision="fp32", img_norm_cfg=norm))
        bb.load_state_dict(sd)
        bb = bb.to(DEV).eval()
        neck = toc3d_amd.build_neck(dict(configs.CPFPN_TINY, precision="fp32"))
        neck.load_state_dict(nsd)
        neck = neck.to(DEV)
        mem = toc3d_amd.TemporalMemory(pseudo_reference_points=minp["pseudo"], **mcfg)
        outs = []
        for f in range(4):
            fr = minp["frames"][f]
            data = {k: v.to(DEV) for k, v in fr["data"].items()}
            mem.pre_update_memory(data)
            q = mem.backbone_queries(Q, prev_exists=data["prev_exists"])
            o = bb(torch.from_numpy(imgs[f]).to(DEV), ego_pose_inv=data["ego_pose_inv"], gumbel_noise=base["gumbel"], **q)
            n0, n1 = neck([o.img_feats["last_feat"]])
            mem.post_update_memory(data, fr["rec_ego_pose"].to(DEV), fr["cls"].to(DEV)[None], fr["bbox"].to(DEV)[None], fr["dec"].to(DEV)[None])
            outs.append((n0.clone(), n1.clone(), [k.clone() for k in o.keep_idx]))
        return outs

    got = run_loop()
    # oracle composition on the host
    ora = MemoryBank(pseudo_reference_points=minp["pseudo"], **mcfg)
    for f in range(4):
        fr = minp["frames"][f]
        ora.pre_update_memory(fr["data"])
        mid = bool(fr["data"]["prev_exists"][0] > 0)
        q = ora.backbone_queries(Q, mid)
        x = torch.from_numpy(I.prepare_images(imgs[f], norm["mean"], norm["std"], norm["to_rgb"], 32))
        with torch.no_grad():
            ref = O.forward_toc3d(sd, cfg, x, q["temp_queries"], q["temp_ref_points"], q["temp_vel"], q["temp_timestamp"], q["temp_ego_pose"],
                                  fr["data"]["ego_pose_inv"], mid, base["gumbel"])
            rn = O.cpfpn(nsd, ref["last_feat"])
        ora.post_update_memory(fr["data"], fr["rec_ego_pose"], fr["cls"], fr["bbox"], fr["dec"])
        e0 = rel_max(got[f][0], rn[0])
        print(f"[loop] frame {f} prev_exists={mid}: neck level-0 rel max err vs oracle {e0:.3e}")
        assert e0 < 1e-3
    again = run_loop()
    for a, b in zip(got, again):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for s in range(3):
            assert torch.equal(a[2][s], b[2][s])


# ---------------------------------------------------------------------------------------------------
# token side of the head (second half)
@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("bf16", 6e-2)])
def test_head_token_embedding_matches_reference_golden(golden_dir, precision, tol):
    """toc3d_amd.HeadTokenEmbedding against tests/golden/head_tokens.npz (the reference's position_embeding + MLN + SELayer_Linear)."""
    from oracle.gen_golden_head import CFG, B, N, H, W
    g = np.load(os.path.join(golden_dir, "head_tokens.npz"))
    m = toc3d_amd.HeadTokenEmbedding(precision=precision, **CFG)
    m.load_state_dict(synth.head_tokens_state_dict(CFG), strict=True)
    m = m.to(DEV).eval()
    inp = synth.head_tokens_inputs(CFG, B, N, H, W)
    memory, pos, cone = m(inp["feats"].to(DEV), inp["intrinsics"].to(DEV), inp["lidar2img"].to(DEV), (H * CFG["stride"], W * CFG["stride"], 3))
    e_cone = rel_max(cone, torch.from_numpy(g["cone"]))
    e_mem, e_pos = rel_max(memory, torch.from_numpy(g["memory"])), rel_max(pos, torch.from_numpy(g["pos_embed"]))
    print(f"[head tokens {precision}] rel max err: cone {e_cone:.2e} memory {e_mem:.2e} pos_embed {e_pos:.2e}")
    assert e_cone < (1e-5 if precision == "fp32" else 1e-5) and e_mem < tol and e_pos < tol
    with pytest.raises(RuntimeError, match="CUDA/HIP"):
        m(inp["feats"], inp["intrinsics"], inp["lidar2img"], (64, 96))


def test_head_token_embedding_full_size_matches_oracle():
    """Shipped sizes (6 views x 20 x 50 tokens, 256 channels, 64 LID depth bins) on the fp32 kernels against the oracle on the host."""
    from oracle import head_tokens_oracle as HO
    cfg = synth.HEAD_TOKENS_CFG
    sd = synth.head_tokens_state_dict(cfg, seed=1)
    inp = synth.head_tokens_inputs(cfg, 1, 6, 20, 50, seed=1)
    m = toc3d_amd.HeadTokenEmbedding(precision="fp32", **cfg)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    memory, pos, _ = m(inp["feats"].to(DEV), inp["intrinsics"].to(DEV), inp["lidar2img"].to(DEV), (320, 800, 3))
    with torch.no_grad():
        rm, rp = HO.token_embeddings(sd, cfg, inp["feats"], inp["intrinsics"], inp["lidar2img"], 320, 800)
    assert tuple(memory.shape) == (1, 6000, 256)
    assert rel_max(memory, rm) < 2e-4 and rel_max(pos, rp) < 2e-4


def test_folded_layernorms_and_riding_prefetch_against_the_explicit_sequence():
    """bf16 dense backbone (no discrete decisions): ffn_ln folded into the w1|w2 / w3 GEMMs (default), norm2 folded into the projection /
    w1|w2 GEMMs as well (default since the end of round 3), against the explicit LayerNorm launches; all within bf16 rounding of each other and of the fp32 oracle.
    The weight prefetch riding on the attention launches must not change a bit."""
    cfg = configs.get("eva_dense")
    sd = synth.make_state_dict(cfg)
    inp = synth.make_inputs(cfg, views_per_frame=1)
    x = inp["x"].to(DEV)
    with torch.no_grad():
        ref = O.forward_eva(sd, cfg, inp["x"])["last_feat"]
    feats = {}
    for tag, ffn, n2, pf in (("explicit", False, False, 0), ("ffn_ln folded", True, False, 0), ("ffn_ln + norm2 folded", True, True, 0),
                             ("ffn_ln folded + prefetch", True, False, 48)):
        m = toc3d_amd.build_backbone(dict(cfg, precision="bf16"))
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV).eval()
        m.fold_ffn_ln, m.fold_norm2, m.prefetch_weights, m.autotune = ffn, n2, pf, False
        for _ in range(3):                                       # eager warm-up, recorded plan, replay
            f = m(x)["last_feat"].clone()
        feats[tag] = f
        print(f"[{tag}] rel l2 vs fp32 oracle {rel_l2(f, ref):.3e}")
        del m
    e0 = rel_l2(feats["explicit"], ref)
    for tag in ("ffn_ln folded", "ffn_ln + norm2 folded"):
        assert rel_l2(feats[tag], ref) < 1.2 * e0 + 2e-3, tag
        assert rel_l2(feats[tag], feats["explicit"]) < 3e-2, tag
    assert torch.equal(feats["ffn_ln folded + prefetch"], feats["ffn_ln folded"])


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16", 2e-2)])
def test_full_size_neck_matches_oracle(precision, tol):
    """CPFPN at the shipped size (1024 -> 256 channels, 6 x 20 x 50 tokens; ToC3D_faster.py:70-74): 1x1 lateral GEMM + the implicit-GEMM 3x3
    conv against the oracle's F.conv2d composition (necks/cp_fpn.py:156-208), NCHW input (copy path) and the backbone's NHWC buffer layout."""
    nsd = synth.neck_state_dict(configs.CPFPN_CFG)
    neck = toc3d_amd.build_neck(dict(configs.CPFPN_CFG, precision=precision))
    neck.load_state_dict(nsd)
    neck = neck.to(DEV).eval()
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(6, 1024, 20, 50, generator=g)
    with torch.no_grad():
        r0, r1 = O.cpfpn(nsd, feat)
    n0, n1 = neck([feat.to(DEV)])
    e0, e1 = rel_max(n0, r0), rel_max(n1, r1)
    print(f"[full-size neck {precision}] rel max err level0 {e0:.3e} level1 {e1:.3e}")
    assert tuple(n0.shape) == (6, 256, 20, 50) and tuple(n1.shape) == (6, 256, 10, 25) and e0 < tol and e1 < tol
    nhwc = feat.permute(0, 2, 3, 1).contiguous().to(DEV)              # what the backbone hands over: an NCHW view of an NHWC buffer
    for _ in range(3):                                                 # eager, recorded, replayed
        m0, _ = neck([nhwc.permute(0, 3, 1, 2)])
    assert torch.equal(m0, n0)


@pytest.mark.parametrize("name,prev", [("toc3d_tiny", True), ("toc3d_tiny", False), ("toc3d_faster", True), ("eva_dense", True)])
def test_fp32x3_on_planes_is_bit_identical_to_the_in_kernel_split(name, prev):
    """precision="fp32x3" ships with its GEMM operands as (hi, lo) bf16 planes (schedule switch x3_planes: weights packed once, activations written as planes by
    their producers; include/toc3d.h TOC3D_DTYPE_F32X3W / F32X3P).  Same arithmetic as the in-LDS split of TOC3D_DTYPE_F32X3: the whole forward returns the
    same bits (eager forward and replayed plan)."""
    cfg = configs.get(name)
    sd = synth.make_state_dict(cfg)
    inp = synth.make_inputs(cfg, views_per_frame=2 if name == "toc3d_tiny" else 6)
    outs = []
    for planes in (False, True):
        m = toc3d_amd.build_backbone(dict(cfg, precision="fp32x3", schedule=dict(x3_planes=planes, x3_attention=False, attn_rot=False)))   # (the attention's own x3 products and its pre-rotated form on planes: next test)
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV).eval()
        assert m.x3_planes is planes
        for _ in range(3):                   # eager, recording, replay
            o = run_toc3d(m, inp, prev) if synth.is_toc3d(cfg) else m(inp["x"].to(DEV))
        outs.append((o.img_feats["last_feat"] if synth.is_toc3d(cfg) else o["last_feat"]).clone())
    assert torch.equal(outs[0], outs[1]), f"max abs difference {(outs[0] - outs[1]).abs().max().item():.3e}"


def test_fp32x3_attention_products_stay_parity_grade():
    """x3_attention (default on for fp32x3): the attention's contractions as bf16 x 3 products instead of exact-f32 MFMAs.  Not bit-identical -- the ViT-L forward
    moves by ~1e-5 relative, far inside the 1e-3 bar the goldens pin (test_vitl_fp32_matches_reference[fp32x3] runs with the default)."""
    cfg = configs.get("toc3d_faster")
    sd = synth.make_state_dict(cfg)
    inp = synth.make_inputs(cfg, views_per_frame=6)
    outs = []
    for x3a, rot in ((False, False), (True, False), (True, True)):
        m = toc3d_amd.build_backbone(dict(cfg, precision="fp32x3", schedule=dict(x3_attention=x3a, attn_rot=rot)))
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV).eval()
        outs.append(run_toc3d(m, inp, True).img_feats["last_feat"].clone())
    e = rel_max(outs[1], outs[0].cpu())
    print(f"[fp32x3] attention with bf16 x 3 products vs exact-f32 attention: rel max diff {e:.3e}")
    assert e < 2e-4
    # round 6, the shipped schedule: RoPE + q scale in the x3 q|k|v epilogue, rows as (hi, lo) planes, toc3d_window_attention_rot on planes (exp2-based online softmax)
    e = rel_max(outs[2], outs[0].cpu())
    print(f"[fp32x3] pre-rotated attention on planes vs exact-f32 attention: rel max diff {e:.3e}")
    assert e < 2e-4
