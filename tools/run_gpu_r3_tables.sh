# tile tables: (1) the (config, size, precision) points that had none, (2) the shipped tables completed with the shapes of a first frame (picks kept)
mkdir -p gpurun_out/tuned2 gpurun_out/tuned3
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
if [ "$1" = "new" ]; then
timeout 2400 python tools/make_tuned_tables.py gpurun_out/tuned2 toc3d_faster:800x1600:bf16 toc3d_fast:320x800:fp32 eva_dense:320x800:fp32 toc3d_fast:320x800:fp32x3 eva_dense:320x800:fp32x3 toc3d_faster:640x1600:fp32x3 2>&1 | grep -v amdgpu.ids | tail -8
fi
TOC3D_KEEP_TABLE=1 timeout 2400 python tools/make_tuned_tables.py gpurun_out/tuned3 toc3d_faster:320x800:bf16:1,2 toc3d_faster:320x800:fp32 toc3d_fast:320x800:bf16 toc3d_faster:640x1600:bf16 toc3d_faster:320x800:fp32x3 toc3d_faster:320x800:fp32x6 2>&1 | grep -v amdgpu.ids | tail -8
for v in 1 0; do
TOC3D_SIDE_LANES=$v timeout 600 python bench.py --no-cpu-baseline --no-batched --no-other-configs --no-parity-path > gpurun_out/r3_side_bench.json 2> gpurun_out/r3_side_bench.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r3_side_bench.json').read().strip().splitlines()[-1])
print('SIDE_LANES=$v', round(d['value'], 2), 'frames/s', [round(x, 3) for x in d['repetitions']['ms_per_step_each']])
PY
done
