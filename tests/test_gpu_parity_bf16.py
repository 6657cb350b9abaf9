"""GPU: parity of the BENCHMARKED precision (bf16 operands, f32 accumulate, f32 residual stream), bounded tightly.

BASELINE.md section 4 / SURVEY.md 7(b,c): a bf16 run of this network has two error sources -- the arithmetic itself, and the
chaos of top-k flips (a token whose score moves by one bf16 ulp changes every later block).  They are separated here:

* **forced selection**: the reference's own image-level scores and soft masks (``stage{s}.score`` / ``token_mask{s}`` of the
  golden fixtures, produced by the REAL reference, oracle/gen_golden.py) replace every scorer stage's output, so every block
  selects exactly the reference's tokens; what is left is arithmetic error, measured against the reference's fp32 features;
* **torch-bf16 control**: the oracle's own code with every nn.Linear / attention matmul in torch bf16 on the same GPU
  (``oracle.toc3d_oracle.contractions("bf16")``), forced and free-running.  The HIP path must be no worse than 1.2x the control.

Per-block budgets: with forced selection the residual stream after every block is compared with the fp32 oracle's, HIP-bf16 next
to the control, block by block.
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
ARGS = ("x", "temp_queries", "temp_ref_points", "temp_vel", "temp_timestamp", "temp_ego_pose", "ego_pose_inv")


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()


def rel_max(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()


def iou(a, b):
    res = []
    for ra, rb in zip(np.asarray(a.cpu()), np.asarray(b)):
        sa, sb = set(ra.tolist()), set(rb.tolist())
        res.append(len(sa & sb) / max(1, len(sa | sb)))
    return min(res)


def build(name, precision, stress=False):
    cfg = configs.get(name)
    m = toc3d_amd.build_backbone(dict(cfg, precision=precision))
    m.load_state_dict(synth.make_state_dict(cfg, stress=stress), strict=True)
    m = m.to(DEV).eval()
    m.autotune = False                    # every tile variant accumulates in the same order: the choice cannot change a bit
    if synth.is_toc3d(cfg):
        from toc3d_amd.testing import instrument
        instrument(m)                     # forced_scores= / block_hook (test instruments, eager launches)
    return cfg, m


def run_hip(m, inp, prev=True, forced=None):
    d = lambda t: t.to(DEV)
    return m(d(inp["x"]), temp_queries=d(inp["temp_queries"]), prev_exists=prev, temp_ref_points=d(inp["temp_ref_points"]),
             temp_vel=d(inp["temp_vel"]), temp_timestamp=d(inp["temp_timestamp"]), temp_ego_pose=d(inp["temp_ego_pose"]),
             ego_pose_inv=d(inp["ego_pose_inv"]), gumbel_noise=inp["gumbel"], forced_scores=forced)


_SD_DEV = {}


def oracle_on_gpu(name, cfg, inp, mode, prev=True, forced=None, capture=None, stress=False):
    """The oracle's code on the GPU: fp32 (checker) or with torch-bf16 contractions (control)."""
    key = name + ("/stress" if stress else "")
    if key not in _SD_DEV:
        _SD_DEV.clear()                   # one ViT-L state dict (1.2 GB) on the device at a time
        _SD_DEV[key] = {k: v.to(DEV) for k, v in synth.make_state_dict(cfg, stress=stress).items()}
    sd = _SD_DEV[key]
    a = [inp[k].to(DEV) for k in ARGS]
    with torch.no_grad(), O.contractions(mode):
        return O.forward_toc3d(sd, cfg, *a, prev, [g.to(DEV) for g in inp["gumbel"]], capture=capture,
                               forced=None if forced is None else [(s.to(DEV), m.to(DEV)) for s, m in forced])


def forced_from_golden(g):
    return [(torch.from_numpy(g[f"stage{s}.score"]), torch.from_numpy(g[f"token_mask{s}"]).reshape(g[f"stage{s}.score"].shape)) for s in range(3)]


def golden_feat(g):
    key = "last_feat.c16" if "last_feat.c16" in g else "last_feat.c32"
    return torch.from_numpy(g[key]), int(key.rsplit("c", 1)[1])


# (fixture, config, hw, prev): BASELINE.json configs 1, 2, 4 (+ the reference's own 1600x800 input and a first frame)
CASES = [("vitl_toc3d_faster", "toc3d_faster", (320, 800), True),
         ("vitl_toc3d_fast", "toc3d_fast", (320, 800), True),
         ("vitl_toc3d_faster_first", "toc3d_faster", (320, 800), False),
         ("vitl_toc3d_faster_1600x640", "toc3d_faster", (640, 1600), True),
         ("vitl_toc3d_faster_1600x800", "toc3d_faster", (800, 1600), True)]


@pytest.mark.parametrize("fixture,name,hw,prev", CASES)
def test_vitl_bf16_forced_selection_within_control(golden_dir, fixture, name, hw, prev):
    """Arithmetic error of the bf16 path with the reference's token selection forced, next to the torch-bf16 control."""
    cfg, m = build(name, "bf16")
    inp = synth.make_inputs(cfg, views_per_frame=6, hw=hw)
    g = np.load(os.path.join(golden_dir, fixture + ".npz"))
    forced = forced_from_golden(g)
    ref, step = golden_feat(g)
    out = run_hip(m, inp, prev, forced)
    feat = out.img_feats["last_feat"]
    for s in range(3):                    # forced scores -> the image-level lists are the reference's, bit for bit
        assert np.array_equal(out.keep_idx[s].cpu().numpy(), g[f"keep_idx{s}"]), f"stage {s}"
    ctl = oracle_on_gpu(name, cfg, inp, "bf16", prev, forced)["last_feat"]
    e_hip, e_ctl = rel_l2(feat[:, ::step], ref), rel_l2(ctl[:, ::step], ref)
    m_hip, m_ctl = rel_max(feat[:, ::step], ref), rel_max(ctl[:, ::step], ref)
    tl2 = (feat.double().norm(dim=1).cpu() - torch.from_numpy(g["last_feat.token_l2"])).abs().max().item() / g["last_feat.token_l2"].max()
    print(f"[{fixture} bf16 forced] rel l2: hip {e_hip:.3e}  torch-bf16 control {e_ctl:.3e}   rel max: hip {m_hip:.3e} control {m_ctl:.3e}"
          f"   token-norm err {tl2:.3e}")
    assert torch.isfinite(feat).all()
    assert e_hip <= 1.2 * e_ctl, "HIP bf16 arithmetic error exceeds 1.2x the torch-bf16 control"
    assert e_hip < 3e-2 and m_hip < 8e-2 and tl2 < 3e-2          # absolute ceilings: measured 2e-2-class rel l2 (random weights)


@pytest.mark.parametrize("fixture,name,hw,prev", CASES)
def test_vitl_bf16_free_running_within_control(golden_dir, fixture, name, hw, prev):
    """Free-running bf16 (what bench.py times): selection decided by the bf16 scores.  Error vs the fp32 reference next to
    the free-running control, and how many kept tokens flipped."""
    cfg, m = build(name, "bf16")
    inp = synth.make_inputs(cfg, views_per_frame=6, hw=hw)
    g = np.load(os.path.join(golden_dir, fixture + ".npz"))
    ref, step = golden_feat(g)
    out = run_hip(m, inp, prev)
    feat = out.img_feats["last_feat"]
    c = oracle_on_gpu(name, cfg, inp, "bf16", prev)
    e_hip, e_ctl = rel_l2(feat[:, ::step], ref), rel_l2(c["last_feat"][:, ::step], ref)
    i_hip = [iou(out.keep_idx[s], g[f"keep_idx{s}"]) for s in range(3)]
    i_ctl = [iou(c["keep_idx"][s], g[f"keep_idx{s}"]) for s in range(3)]
    print(f"[{fixture} bf16 free] rel l2: hip {e_hip:.3e} control {e_ctl:.3e}   keep IoU: hip {np.round(i_hip, 4).tolist()} control {np.round(i_ctl, 4).tolist()}")
    assert torch.isfinite(feat).all()
    # top-k flips are chaotic (one flipped token moves every later block), so two free-running bf16 runs are only comparable as a
    # band.  Measured on MI355X (r02): rel l2 hip / control 0.118 / 0.123 (faster), 0.120 / 0.117 (fast), 0.137 / 0.163 (first frame);
    # worst-view stage-3 IoU 0.936 / 0.936, 0.961 / 0.942, 0.840 / 0.887.  Stage 1 is decided by six dense blocks only: near 0.99.
    # Round 3: also both 1600-wide inputs (BASELINE.json config 4 and the reference's own 1600 x 800).
    assert e_hip <= 1.2 * e_ctl + 5e-3
    assert min(i_hip) >= min(i_ctl) - 0.06 and i_hip[0] > 0.98


def test_vitl_stress_fixture_heavy_tailed_activations(golden_dir):
    """VERDICT r05 item 7 / "missing 1": every other fixture draws N(0, sigma^2) weights, whose activations have no outliers; trained ViTs do (a handful of residual-stream
    channels tens of times wider than the rest, LayerNorm gains that single them out, large-norm tokens).  `vitl_toc3d_faster_stress.npz` is the REAL reference's fp32 output
    (oracle/gen_golden.py vitl_stress) on weights and inputs with that pattern planted (toc3d_amd.synth.STRESS_CHANNELS: output channel abs-max 1530 against a median of 147).
    Reported and bounded on it: the exact-f32 path, the parity-grade default (fp32x3), and bf16 -- forced selection and free running, next to the torch-bf16 control."""
    name = "toc3d_faster"
    g = np.load(os.path.join(golden_dir, "vitl_toc3d_faster_stress.npz"))
    ref, step = golden_feat(g)
    forced = forced_from_golden(g)
    cfg = configs.get(name)
    inp = synth.make_inputs(cfg, views_per_frame=6, stress=True)
    rows = {}
    for precision in ("fp32", "fp32x6", "fp32x3", "bf16"):
        _, m = build(name, precision, stress=True)
        out = run_hip(m, inp, True)
        feat = out.img_feats["last_feat"]
        assert torch.isfinite(feat).all()
        off = ((feat[:, ::step].cpu() - ref).abs().amax(dim=1) > 1e-3 * ref.abs().max()).float().mean().item()       # share of tokens off by > 1e-3 (free running)
        rows[precision] = dict(free_max=rel_max(feat[:, ::step], ref), free_l2=rel_l2(feat[:, ::step], ref), free_tokens_off=off,
                               iou=[iou(out.keep_idx[s], g[f"keep_idx{s}"]) for s in range(3)])
        fo = run_hip(m, inp, True, forced).img_feats["last_feat"]
        rows[precision].update(forced_max=rel_max(fo[:, ::step], ref), forced_l2=rel_l2(fo[:, ::step], ref))
        del m
        torch.cuda.empty_cache()
    cf = oracle_on_gpu(name, cfg, inp, "bf16", True, forced, stress=True)["last_feat"]
    cr = oracle_on_gpu(name, cfg, inp, "bf16", True, stress=True)
    ctl = dict(forced_l2=rel_l2(cf[:, ::step], ref), forced_max=rel_max(cf[:, ::step], ref), free_l2=rel_l2(cr["last_feat"][:, ::step], ref),
               iou=[iou(cr["keep_idx"][s], g[f"keep_idx{s}"]) for s in range(3)])
    for k, v in list(rows.items()) + [("torch-bf16 control", ctl)]:
        print(f"[stress fixture {k}] " + "  ".join(f"{a} {b:.3e}" if not isinstance(b, list) else f"{a} {np.round(b, 4).tolist()}" for a, b in v.items()))
    # Measured (round 6, MI355X): fp32 forced / free 9.9e-5 (kept lists equal); fp32x3 forced 7.7e-4 -- 25x its 3e-5 on the benign fixtures: the outlier channels carry
    # the hi.lo + lo.hi cross terms at 20x the magnitude of everything else, STILL inside 1e-3 --, free running one window-level near-tie breaks the other way (1.8e-2 on
    # that window's tokens, rel. L2 2.5e-3, image-level lists equal); bf16 forced 4.9e-2 rel. L2 against 6.8e-2 for the torch-bf16 control.
    # The f32-buffer paths keep the 1e-3 bar on the ARITHMETIC (selection forced); free running, exact fp32 must also meet it, the product forms are bounded like at the
    # 1600-wide inputs: at most two windows' worth of tokens may sit on the other side of a near-tie.
    allowed = 2.2 * 400 / 6000
    for precision in ("fp32", "fp32x6", "fp32x3"):
        r = rows[precision]
        assert r["forced_max"] < 1e-3, (precision, r)
        assert min(r["iou"]) > 0.99, (precision, r)
        if precision == "fp32":
            assert r["free_max"] < 1e-3, (precision, r)
        else:
            assert r["free_tokens_off"] < allowed and r["free_l2"] < 1e-2, (precision, r)
    # bf16: arithmetic error with the selection forced within 1.2x the torch-bf16 control, free running inside the control's band
    b = rows["bf16"]
    assert b["forced_l2"] <= 1.2 * ctl["forced_l2"] + 1e-3, (b, ctl)
    assert b["free_l2"] <= 1.2 * ctl["free_l2"] + 5e-3 and min(b["iou"]) >= min(ctl["iou"]) - 0.06, (b, ctl)


@pytest.mark.parametrize("name", ["toc3d_faster", "toc3d_fast"])
def test_vitl_bf16_per_block_error_budget(golden_dir, name):
    """Residual stream after every block, forced selection: HIP bf16 vs the fp32 oracle, block by block, next to the control."""
    cfg, m = build(name, "bf16")
    inp = synth.make_inputs(cfg, views_per_frame=6)
    g = np.load(os.path.join(golden_dir, f"vitl_{name}.npz"))
    forced = forced_from_golden(g)
    snaps = {}
    m.block_hook = lambda i, gp, carried: None if carried else snaps.__setitem__(i, gp["x"][:, ::16].clone())
    run_hip(m, inp, True, forced)
    m.block_hook = None
    cap32, cap16 = {}, {}
    oracle_on_gpu(name, cfg, inp, "fp32", True, forced, capture=cap32)
    ref = {i: cap32[f"block{i}.out"].reshape(-1, cfg["embed_dim"])[:, ::16].clone() for i in snaps}
    del cap32
    oracle_on_gpu(name, cfg, inp, "bf16", True, forced, capture=cap16)
    rows = []
    for i in sorted(snaps):
        c = cap16[f"block{i}.out"].reshape(-1, cfg["embed_dim"])[:, ::16]
        rows.append((i, rel_l2(snaps[i], ref[i]), rel_l2(c, ref[i])))
    print(f"[{name} bf16 forced] per-block rel l2 (block: hip / control): " + "  ".join(f"{i}: {a:.2e}/{b:.2e}" for i, a, b in rows))
    # the reference golden pins the oracle's per-block values where the fixture holds them (blocks 5, 6, 11, 17)
    for i in (5, 6, 11, 17):
        if f"block{i}.out.c16" in g and i in ref:
            assert rel_max(ref[i].reshape(g[f"block{i}.out.c16"].shape), torch.from_numpy(g[f"block{i}.out.c16"])) < 1e-4
    assert len(rows) >= 15
    for i, e_hip, e_ctl in rows:
        assert e_hip <= 1.2 * e_ctl + 2e-4, f"block {i}: hip {e_hip:.3e} vs control {e_ctl:.3e}"
    assert rows[-1][1] < 3e-2
    assert all(b[1] <= 1.6 * a[1] + 1e-3 for a, b in zip(rows, rows[1:])), "error must grow smoothly along the depth (no broken block)"


def window_scores(score, h, w, L):
    """image-level scores [V, h*w] -> [windows, L*L] with the reference's pad value (toc3d_eva_vit.py:412-415, eva_utils.py:89-110)."""
    V = score.shape[0]
    s = score.reshape(V, h, w)
    ph, pw = (L - h % L) % L, (L - w % L) % L
    s = torch.nn.functional.pad(s, (0, pw, 0, ph), value=-1e6)
    return s.reshape(V, (h + ph) // L, L, (w + pw) // L, L).permute(0, 1, 3, 2, 4).reshape(-1, L * L)


@pytest.mark.parametrize("precision", ["fp32", "fp32x3", "fp32x6"])
@pytest.mark.parametrize("fixture,hw", [("vitl_toc3d_faster_1600x640", (640, 1600)), ("vitl_toc3d_faster_1600x800", (800, 1600))])
def test_vitl_1600_fp32_matches_reference(golden_dir, fixture, hw, precision):
    """BASELINE.json config 4 (6 x 1600x640) and the reference's own hi-res input (6 x 1600x800, ToC3D_faster_1600.py:43,177), all six
    views, strict-parity path (and the bf16 x 3 parity-grade fast path) against the REAL reference's golden."""
    cfg, m = build("toc3d_faster", precision)
    inp = synth.make_inputs(cfg, views_per_frame=6, hw=hw)
    g = np.load(os.path.join(golden_dir, fixture + ".npz"))
    ref, step = golden_feat(g)
    out = run_hip(m, inp, True)
    feat = out.img_feats["last_feat"]
    assert tuple(feat.shape) == (6, 1024, hw[0] // 16, hw[1] // 16)
    ious = [iou(out.keep_idx[s], g[f"keep_idx{s}"]) for s in range(3)]
    flips = []
    for s in range(3):
        assert ious[s] > 0.99
        md = (out.token_masks[s][..., 0].cpu() - torch.from_numpy(g[f"token_mask{s}"])).abs()
        flips.append((md > 5e-3).float().mean().item())
    # the exact-f32 path reproduces every mask; the product forms may break ONE near-tie the other way (below), seen on either hi-res input
    # depending on the product form (r03: x3 / x6 at 1600x800, x6 at 1600x640) -- bounded, never silently accepted for the exact path
    one_flip = precision != "fp32" and (max(flips) > 1e-9 or min(ious) < 1.0)      # a soft-mask flip, or a per-window top-k flip (kept-set IoU < 1)
    assert max(flips) < (2e-2 if one_flip else 1e-9), flips
    err = rel_max(feat[:, ::step], ref)
    tl2 = (feat.double().norm(dim=1).cpu() - torch.from_numpy(g["last_feat.token_l2"])).abs().max().item() / g["last_feat.token_l2"].max()
    bad = ((feat[:, ::step].cpu() - ref).abs().amax(dim=1) > 1e-3 * ref.abs().max()).float().mean().item()
    print(f"[{fixture} {precision}] rel max err {err:.3e} token-norm err {tl2:.3e}  kept-set IoU {[round(v, 5) for v in ious]}  tokens off by > 1e-3: {100 * bad:.4f} %")
    # Round 6: the f64 run of the REAL reference on the same inputs (oracle/gen_golden_f64.py) as the arbiter of every per-window selection.  What it established: the
    # smallest kept/dropped gap of a window score is 1e-6 .. 7e-6 at these inputs (near_ties), the reference's own fp32 CPU forward stands 3e-6 from its f64 run and
    # happens to make every window's f64 selection.  Here: a window whose kept set differs from the f64 run's must be a near-tie -- its f64 gap below twice the error
    # of OUR image-level scores against the f64 scores -- and there are at most two such windows per (stage, window side).  A flip anywhere else fails.
    a64 = np.load(os.path.join(golden_dir, fixture + "_f64.npz"))
    plan = next(iter(m._plans.values()))
    hh, ww = hw[0] // 16, hw[1] // 16
    flipped = []
    for s in range(3):
        ours = plan["score"][s].view(6, -1).cpu().double()
        ref64 = torch.from_numpy(a64[f"stage{s}.score"])
        serr = (ours - ref64).abs().max().item()
        for L in (16, 20):
            k = int(L * L * cfg["token_ratio"][s])
            wo, w6 = window_scores(ours, hh, ww, L), window_scores(ref64, hh, ww, L)
            ko = torch.sort(wo, dim=1, descending=True, stable=True)[1][:, :k].sort(dim=1)[0]
            o6 = torch.sort(w6, dim=1, descending=True, stable=True)
            k6 = o6[1][:, :k].sort(dim=1)[0]
            diff = (ko != k6).any(dim=1).nonzero().flatten().tolist()
            for wi in diff:
                gap = (o6[0][wi, k - 1] - o6[0][wi, k]).item()
                flipped.append((s, L, wi, gap, serr))
                assert gap < 2 * serr + 1e-12, f"stage {s} side {L} window {wi}: kept set differs from the f64 reference's at a gap of {gap:.2e}, score error {serr:.2e}"
            assert len(diff) <= 2, (s, L, diff)
    print(f"[{fixture} {precision}] windows whose kept set differs from the f64 reference's (stage, side, window, f64 gap, our score error): {flipped}")
    if precision == "fp32":
        assert not flipped
    if one_flip:
        # This input (30 000 tokens) holds a per-window top-k near-tie below 1e-6: the exact-f32 kernels happen to break it the way the reference's
        # CPU summation order does, the bf16 x 3 (3e-5) and even the f32-grade bf16 x 6 (6e-6 everywhere else) forms break it the other way --
        # measured r03, identically for both: ONE 20x20 window (1.3 % of the tokens) differs, rel max err 0.15 on its tokens, everything else at
        # the path's own error.  The chaos BASELINE.md section 4 describes, at the smallest possible scale; stated, not hidden: the bound
        # here is the flip count (one window), not 1e-3.
        # one flipped selection touches one 20 x 20 global window (400 tokens) directly and a few neighbours through the last blocks: at most 2.2 windows' worth
        # of the 6 x h x w tokens (1600x800: 2.9 %, 1600x640: 3.7 %; measured 1.3 % / 2.6 %)
        allowed = 2.2 * 400 / (6 * (hw[0] // 16) * (hw[1] // 16))
        assert bad < allowed and rel_l2(feat[:, ::step], ref) < 4e-2
        ok = (feat[:, ::step].cpu() - ref).abs().amax(dim=1) <= 1e-3 * ref.abs().max()
        assert ok.float().mean().item() > 1.0 - allowed
    else:
        assert err < 1e-3 and tl2 < 1e-3


@pytest.mark.parametrize("precision", ["fp32", "fp32x3", "fp32x6"])
def test_vitl_first_frame_fp32_matches_reference(golden_dir, precision):
    """prev_exists=False at full size: ScoreBasedTokenSelector.score (toc3d_utils.py:114-129) feeds the selection."""
    cfg, m = build("toc3d_faster", precision)
    inp = synth.make_inputs(cfg, views_per_frame=6)
    g = np.load(os.path.join(golden_dir, "vitl_toc3d_faster_first.npz"))
    ref, step = golden_feat(g)
    out = run_hip(m, inp, False)
    feat = out.img_feats["last_feat"]
    for s in range(3):
        assert iou(out.keep_idx[s], g[f"keep_idx{s}"]) > 0.99
    err = rel_max(feat[:, ::step], ref)
    print(f"[vitl toc3d_faster first frame fp32] rel max err {err:.3e}")
    assert err < 1e-3


def test_forced_selection_reproduces_free_running_on_the_parity_path(golden_dir):
    """fp32: forcing the reference's scores must give the same features as running free (the fp32 scores agree to 1e-6)."""
    cfg, m = build("toc3d_tiny", "fp32")
    inp = synth.make_inputs(cfg, views_per_frame=2)
    g = np.load(os.path.join(golden_dir, "tiny_toc3d_prev.npz"))
    forced = [(torch.from_numpy(g[f"stage{s}.score"]).reshape(2, -1), torch.from_numpy(g[f"token_mask{s}"]).reshape(2, -1)) for s in range(3)]
    a = run_hip(m, inp, True, forced).img_feats["last_feat"].clone()
    b = run_hip(m, inp, True).img_feats["last_feat"].clone()
    ref = torch.from_numpy(g["last_feat"])
    assert rel_max(a, ref) < 1e-3 and rel_max(b, ref) < 1e-3 and rel_max(a, b) < 1e-4


def _tuned_tables():
    import glob
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "toc3d_amd", "tuned")
    out = []
    for p in sorted(glob.glob(os.path.join(root, "*.json"))):
        m = re.match(r"(.+)_(\d+)x(\d+)_(bf16|fp32|fp32x3|fp32x6)\.json$", os.path.basename(p))
        out.append((p, m.group(1), (int(m.group(2)), int(m.group(3))), m.group(4)))
    return out


@pytest.mark.parametrize("path,name,hw,precision", _tuned_tables(), ids=lambda v: os.path.basename(v) if isinstance(v, str) and v.endswith(".json") else None)
def test_shipped_tile_tables_give_the_default_variants_bits(path, name, hw, precision):
    """The parity tests above run the default tile variants (autotune off) and rely on "every variant accumulates in the same order".
    This closes the loop for what bench.py actually launches: for EVERY shipped (config, size, precision) table, the full-size forward with
    the table loaded is bit-identical to the forward with the default variants -- features, masks and kept-token lists."""
    import json
    cfg, m = build(name, precision)
    inp = synth.make_inputs(cfg, views_per_frame=6, hw=hw)
    toc = synth.is_toc3d(cfg)

    def fwd():
        if toc:
            o = run_hip(m, inp, True)
            return [o.img_feats["last_feat"].clone()] + [t.clone() for t in o.token_masks] + [t.clone() for t in o.keep_idx]
        return [m(inp["x"].to(DEV))["last_feat"].clone()]
    m.launch_mode = "eager"
    base = fwd()
    assert m._tuned and all(v == 0 for v in m._tuned.values()), "autotune is off: the default variants ran"
    seen = set(m._tuned)
    m.load_tuning(path)
    table = {tuple(k): v for k, v in json.load(open(path))["table"]}
    assert table and all(m._tuned[k] == v for k, v in table.items())
    # (the rotating q|k|v launches look their tile up under the bias epilogue's key without recording a key of their own; the dense backbone launches four or five other shapes; tables may also hold the neck's shapes and shapes of forms no longer launched)
    assert len(seen & set(table)) >= min(8, len(seen) - 1) and any(table[k] != 0 for k in seen & set(table)), "the table must cover the shapes this forward launches"
    m._plans = {}
    tuned = fwd()
    for a, b in zip(base, tuned):
        assert torch.equal(a, b), f"{os.path.basename(path)}: a shipped tile variant changes the result"
    assert bool(torch.isfinite(base[0]).all())
