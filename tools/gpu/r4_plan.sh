export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_plan.py tests/test_gpu_e2e.py tests/test_gpu_dist.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -6
timeout 1500 python bench.py --no-cpu-baseline > gpurun_out/r4_bench2.json 2> gpurun_out/r4_bench2.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4_bench2.json').read().strip().splitlines()[-1])
print(round(d['value'], 1), 'frames/s', d['repetitions']['ms_per_step_each'], 'cal', round(d['calibration']['gemm_yardstick_tflops']), round(d['calibration']['copy_gb_s']))
r = d['roofline']; print('roofline', {k: r[k] for k in ('frac', 'frac_issued', 'avg_launch_ms', 'event_pair_cost_ms')})
print('ab', d['ab_norm2_fold']['shipped']['median'], d['ab_norm2_fold']['other']['median'], d['ab_norm2_fold']['shipped_over_other'])
print('batched', d.get('batched', {}).get('value'), 'parity', d.get('parity_path', {}).get('value'), 'fast', (d.get('parity_path_fast') or {}).get('value'))
print('other', [(o['config'][:20], round(o['value'], 1)) for o in d.get('other_configs', [])])
PY
