"""GPU: recorded launch plans (toc3d_plan_*, toc3d_amd/plan.py) replay exactly what eager launching computes.

One frame's launch sequence is recorded once per (config, input shape, variant) and replayed from C with one call per frame --
on HIP streams with event edges ("plan") or as an explicitly constructed hipGraph ("graph").  Every replay must return the
bits of the eager path, for changing inputs (the staging copies and the directly launched im2col pick up the new frame), for
several view groups (concurrent lanes), across scorer variants (first frame / later frames) and for the neck.
"""
import pytest
import torch

import toc3d_amd
from toc3d_amd import configs, lib, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(name, precision="bf16", mode="eager", groups=1):
    cfg = configs.get(name)
    m = toc3d_amd.build_backbone(dict(cfg, precision=precision))
    m.load_state_dict(synth.make_state_dict(cfg), strict=True)
    m = m.to(DEV).eval()
    m.autotune = False
    m.launch_mode, m.view_groups = mode, groups
    return cfg, m


def run(m, cfg, inp, prev=True):
    d = lambda t: t.to(DEV)
    if not synth.is_toc3d(cfg):
        return (m(d(inp["x"]))["last_feat"].clone(),)
    o = m(d(inp["x"]), temp_queries=d(inp["temp_queries"]), prev_exists=prev, temp_ref_points=d(inp["temp_ref_points"]),
          temp_vel=d(inp["temp_vel"]), temp_timestamp=d(inp["temp_timestamp"]), temp_ego_pose=d(inp["temp_ego_pose"]),
          ego_pose_inv=d(inp["ego_pose_inv"]), gumbel_noise=inp["gumbel"])
    return (o.img_feats["last_feat"].clone(), *[k.clone() for k in o.keep_idx], *[k.clone() for k in o.drop_idx], *[t.clone() for t in o.token_masks])


def same(a, b):
    return len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("name,vpf", [("toc3d_tiny", 2), ("toc3d_tiny", 6), ("eva_tiny", 2)])
@pytest.mark.parametrize("groups", [1, 2])
@pytest.mark.parametrize("mode", ["plan", "graph"])
def test_replayed_frames_equal_eager_frames(name, vpf, groups, mode):
    cfg, eager = build(name, groups=groups)
    _, replay = build(name, mode=mode, groups=groups)
    frames = [synth.make_inputs(cfg, views_per_frame=vpf, seed=s, epoch_timestamps=(s == 2)) for s in range(4)]
    # frame order 0 1 2 3 1 0: warm-up (eager), recording, replays on new inputs, replays on inputs seen before
    for k in (0, 1, 2, 3, 1, 0):
        a, b = run(eager, cfg, frames[k]), run(replay, cfg, frames[k])
        torch.cuda.synchronize()
        assert same(a, b), f"frame {k} differs between eager and {mode}"
    st = next(iter(replay._plans.values()))["launch"]
    assert all(v.get("cplan") is not None and v["cplan"].num_launches > 20 for v in st.values())


def test_scorer_variants_get_their_own_recordings():
    """First frame of a scene (prev_exists=False), later frames with f32 and with f64 timestamps: three launch sequences."""
    cfg, eager = build("toc3d_tiny")
    _, replay = build("toc3d_tiny", mode="plan", groups=2)
    eager.view_groups = 2
    base = synth.make_inputs(cfg, views_per_frame=2)
    f32ts = dict(base, temp_timestamp=base["temp_timestamp"].float())
    seq = [(base, False), (base, True), (f32ts, True)] * 3
    for inp, prev in seq:
        assert same(run(eager, cfg, inp, prev), run(replay, cfg, inp, prev))
    st = next(iter(replay._plans.values()))["launch"]
    # (prev_exists, timestamp staging buffer, Gumbel noise drawn on the device): the injected-noise runs of this test record the three below
    assert set(st) == {(False, "ts32", False), (True, "ts64", False), (True, "ts32", False)} and all(v["cplan"] is not None for v in st.values())


@pytest.mark.parametrize("mode", ["plan", "graph"])
def test_vitl_replay_is_bit_identical_and_draws_its_own_gumbel_noise(mode):
    cfg, eager = build("toc3d_faster", groups=2)
    _, replay = build("toc3d_faster", mode=mode, groups=2)
    inp = synth.make_inputs(cfg, views_per_frame=6)
    for _ in range(4):
        assert same(run(eager, cfg, inp), run(replay, cfg, inp))
    # without injected noise the soft masks of stages 1, 2 are stochastic (toc3d_utils.py:147): replays must not freeze the noise
    d = lambda t: t.to(DEV)
    kw = dict(temp_queries=d(inp["temp_queries"]), prev_exists=True, temp_ref_points=d(inp["temp_ref_points"]), temp_vel=d(inp["temp_vel"]),
              temp_timestamp=d(inp["temp_timestamp"]), temp_ego_pose=d(inp["temp_ego_pose"]), ego_pose_inv=d(inp["ego_pose_inv"]))
    m1 = replay(d(inp["x"]), **kw).token_masks[0].clone()
    m2 = replay(d(inp["x"]), **kw).token_masks[0].clone()
    assert not torch.equal(m1, m2)


def test_new_weights_invalidate_recordings():
    cfg, m = build("eva_tiny", mode="plan")
    inp = synth.make_inputs(cfg, views_per_frame=2)
    for _ in range(3):
        a = run(m, cfg, inp)
    sd2 = synth.make_state_dict(cfg, seed=1)
    m.load_state_dict(sd2)
    _, fresh = build("eva_tiny")
    fresh.load_state_dict(sd2)
    for _ in range(3):
        b = run(m, cfg, inp)
    assert not same(a, b) and same(b, run(fresh, cfg, inp))


@pytest.mark.parametrize("mode", ["plan", "graph"])
def test_neck_replay_equals_eager(mode):
    cfg, m = build("eva_tiny", mode=mode)
    m.alias_outputs = True                               # the neck reads the backbone's buffer in place: recordable
    necks = []
    for nm in ("eager", mode):
        n = toc3d_amd.build_neck(dict(configs.CPFPN_TINY, precision="bf16"))
        n.load_state_dict(synth.neck_state_dict(configs.CPFPN_TINY))
        n = n.to(DEV)
        n.launch_mode = nm
        necks.append(n)
    for s in range(4):
        inp = synth.make_inputs(cfg, views_per_frame=2, seed=s)
        feat = m(inp["x"].to(DEV))["last_feat"]
        a, b = necks[0]([feat]), necks[1]([feat])
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert any(v.get("cplan") is not None for ws in necks[1]._ws.values() for v in ws["launch"].values())


def test_recording_refuses_real_streams():
    """A call that passes a real stream while a plan is being recorded is an error at toc3d_plan_end, not a silent launch."""
    from toc3d_amd.plan import LaunchPlan
    p = LaunchPlan()
    lib.call("toc3d_plan_begin", p.handle)
    x = torch.zeros(1024, device=DEV)
    y = torch.zeros(1024, device=DEV)
    lib.call("toc3d_copy_bytes", y, x, 4096, torch.cuda.current_stream().cuda_stream)
    with pytest.raises(RuntimeError, match="lane handle"):
        lib.call("toc3d_plan_end", p.handle, 0)


def test_device_side_gumbel_draw_is_part_of_the_plan_and_has_gumbel_statistics():
    """Production calls inject no noise: the draw (toc3d_utils.py:145-147, -log of Exp(1) samples) is a kernel INSIDE the recorded frame, keyed by a
    device-side frame counter -- every replay draws fresh noise, the same (seed, frame) reproduces, and the samples follow Gumbel(0, 1)."""
    import math
    n = 1 << 20
    out = torch.empty(n, device=DEV)
    state = torch.zeros(2, dtype=torch.int64, device=DEV)
    lib.call("toc3d_gumbel_noise", out, n, 1234, state, lib.stream_ptr())
    a = out.clone()
    assert state.tolist() == [1, 0]                                   # the launch advanced the frame counter and reset its ticket
    lib.call("toc3d_gumbel_noise", out, n, 1234, state, lib.stream_ptr())
    b = out.clone()
    assert state.tolist() == [2, 0] and not torch.equal(a, b)
    state.zero_()
    lib.call("toc3d_gumbel_noise", out, n, 1234, state, lib.stream_ptr())
    assert torch.equal(out, a), "same (seed, frame counter) -> same noise"
    state.zero_()
    lib.call("toc3d_gumbel_noise", out, n, 99, state, lib.stream_ptr())
    assert not torch.equal(out, a)
    for x in (a, b):
        x = x.double().cpu()
        assert bool(torch.isfinite(x).all())
        assert abs(x.mean().item() - 0.5772156649) < 5e-3 and abs(x.var().item() - math.pi ** 2 / 6) < 1.5e-2
        xs, _ = torch.sort(x)
        cdf = torch.exp(-torch.exp(-xs))                               # Gumbel(0, 1)
        ks = (cdf - (torch.arange(n, dtype=torch.float64) + 0.5) / n).abs().max().item()
        assert ks < 3e-3, ks                                           # Kolmogorov-Smirnov distance at n = 2^20: ~1.4e-3 at the 5 % level
    assert abs(torch.corrcoef(torch.stack([a[:-1], a[1:]]))[0, 1].item()) < 5e-3
    # inside the model: un-injected calls replay the recorded frame and still see new masks every frame
    cfg = configs.get("toc3d_tiny")
    m = toc3d_amd.build_backbone(dict(cfg, precision="fp32"))
    m.load_state_dict(synth.make_state_dict(cfg))
    m = m.to(DEV).eval()
    inp = synth.make_inputs(cfg, views_per_frame=2)
    d = lambda t: t.to(DEV)
    masks = []
    for _ in range(5):
        o = m(d(inp["x"]), temp_queries=d(inp["temp_queries"]), prev_exists=True, temp_ref_points=d(inp["temp_ref_points"]), temp_vel=d(inp["temp_vel"]),
              temp_timestamp=d(inp["temp_timestamp"]), temp_ego_pose=d(inp["temp_ego_pose"]), ego_pose_inv=d(inp["ego_pose_inv"]))
        masks.append(o.token_masks[0].clone())
    key = next(iter(m._plans))
    st = m._plans[key]["launch"][(True, "ts64" if inp["temp_timestamp"].dtype == torch.float64 else "ts32", True)]
    assert st.get("cplan") is not None, "frames 3-5 were replayed from the recorded plan"
    assert all(not torch.equal(masks[i], masks[j]) for i in range(5) for j in range(i))
    assert int(m._plans[key]["stage"]["rng"][0].item()) == 5
    # ONE frame counter per model: a second input shape (its own plan) continues the stream instead of replaying it from frame 0
    inp3 = synth.make_inputs(cfg, views_per_frame=1)
    m(d(inp3["x"]), temp_queries=d(inp3["temp_queries"]), prev_exists=True, temp_ref_points=d(inp3["temp_ref_points"]), temp_vel=d(inp3["temp_vel"]),
      temp_timestamp=d(inp3["temp_timestamp"]), temp_ego_pose=d(inp3["temp_ego_pose"]), ego_pose_inv=d(inp3["ego_pose_inv"]))
    assert len(m._plans) == 2 and int(m._gumbel_rng[0].item()) == 6
    assert all(pl["stage"]["rng"].data_ptr() == m._gumbel_rng.data_ptr() for pl in m._plans.values())


def test_gumbel_map_is_finite_for_every_32_bit_word():
    """The uniform behind -log(-log(U)) must stay strictly inside (0, 1) for every Philox word: U = 1 gives +inf noise, inf - inf = NaN in the
    soft mask (toc3d_utils.py:147) and a NaN score in the next stage.  The extreme words, every exponent boundary of the 23-bit mantissa path and a
    random sample, against the same map in f64."""
    import numpy as np
    edge = [0, 1, 0x1FF, 0x200, 0x7FFFFFFF, 0x80000000, 0xFFFFFE00, 0xFFFFFDFF, 0xFFFFFFFE, 0xFFFFFFFF]
    rnd = np.random.default_rng(5).integers(0, 1 << 32, size=1 << 16, dtype=np.uint64)
    words = np.concatenate([np.array(edge, dtype=np.uint64), rnd]).astype(np.uint32)
    bits = torch.from_numpy(words.view(np.int32)).to(DEV)
    out = torch.empty(len(words), device=DEV)
    lib.call("toc3d_gumbel_from_bits", bits, len(words), out, lib.stream_ptr())
    torch.cuda.synchronize()
    got = out.double().cpu().numpy()
    assert np.isfinite(got).all()
    u = ((words >> 9).astype(np.float64) + 0.5) * 2.0 ** -23
    assert u.min() >= 2.0 ** -24 and u.max() <= 1 - 2.0 ** -24
    ref = -np.log(-np.log(u))
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    assert got[9] > 15.9 and got[0] < -2.7                               # the largest / smallest sample the map can produce: ~16.6 / ~-2.8
