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
    assert set(st) == {(False, "ts32"), (True, "ts64"), (True, "ts32")} and all(v["cplan"] is not None for v in st.values())


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
