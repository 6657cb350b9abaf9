mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -x -k "scatter_folded or carried or tiny" 2>&1 | tail -8
for rep in 1 2; do
for v in 0 1; do
TOC3D_FUSE_SCATTER=$v timeout 600 python bench.py --no-cpu-baseline --no-batched --no-other-configs --no-parity-path > gpurun_out/r3_fuse_bench.json 2> gpurun_out/r3_fuse_bench.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r3_fuse_bench.json').read().strip().splitlines()[-1])
print('FUSE_SCATTER=$v', round(d['value'], 2), 'frames/s', [round(x, 3) for x in d['repetitions']['ms_per_step_each']], d['roofline'].get('hbm_kernels', {}).get('gather_merge_ln_ex'))
PY
done
done
