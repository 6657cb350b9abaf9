export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ops.py -m gpu -q -x --tb=short -p no:cacheprovider -k "planes or (fp32x3 and vitl and faster) or packed" 2>&1 | tail -8
for P in 0 1; do
X3P=$P timeout 600 python - <<'PY'
import os, torch, toc3d_amd
from toc3d_amd import configs, synth
from toc3d_amd import dist as tdist
planes = os.environ["X3P"] == "1"
cfg = configs.get("toc3d_faster"); sd = synth.make_state_dict(cfg)
dev = torch.device("cuda", 0)
m = toc3d_amd.build_backbone(dict(cfg, precision="fp32x3", schedule=dict(x3_planes=planes))); m.load_state_dict(sd); m = m.to(dev).eval(); m.alias_outputs = True
t = "toc3d_amd/tuned/toc3d_faster_320x800_fp32x3.json"
m.load_tuning(t)
inp = synth.make_inputs(cfg, views_per_frame=6)
kw = {k: inp[k].to(dev) for k in ("temp_queries", "temp_ref_points", "temp_vel", "temp_timestamp", "temp_ego_pose", "ego_pose_inv")}
g = [x.to(dev) for x in inp["gumbel"]]; x = inp["x"].to(dev)
step = lambda: m(x, prev_exists=True, gumbel_noise=g, **kw)
for _ in range(4): step()
ts = [tdist.timed_steps(step, 20, 1, dev) for _ in range(3)]
print(f"fp32x3 backbone, x3_planes={planes}: " + " ".join(f"{20 / t:.1f}" for t in ts) + " frames/s", flush=True)
PY
done 2>&1 | grep -v amdgpu | tee gpurun_out/x3_planes_fps.txt
