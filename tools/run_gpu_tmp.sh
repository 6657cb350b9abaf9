cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity_bf16.py -m gpu -q --tb=short -p no:cacheprovider -k "fp32x6 or shipped_tile" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_attn_rot.py tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -k "not fp32" 2>&1 | tail -3
for rep in 1 2; do
for v in 0 1; do
TOC3D_BIG_WINDOWS_FIRST=$v timeout 600 python bench.py --no-cpu-baseline --no-batched --no-parity-path > gpurun_out/r3_bwf_bench.json 2> gpurun_out/r3_bwf_bench.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r3_bwf_bench.json').read().strip().splitlines()[-1])
print('BIG_WINDOWS_FIRST=$v', round(d['value'], 2), 'frames/s', [round(x, 3) for x in d['repetitions']['ms_per_step_each']], 'other', [round(o['value'], 1) for o in d.get('other_configs', [])])
PY
done
done
