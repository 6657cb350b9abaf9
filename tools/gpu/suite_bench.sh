# default bench line, whole GPU suite, smoke
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
T0=$(date +%s); timeout 1500 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "default bench.py wall time: $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print(round(d['value'], 1), 'frames/s', d['repetitions']['ms_per_step_each'])
print('calibration', d.get('calibration'))
r = d['roofline']; print('roofline', {k: r[k] for k in ('frac', 'frac_issued', 'frac_event_timed', 'avg_launch_ms', 'avg_launch_ms_event_timed', 'event_pair_cost_ms', 'traffic')})
print('hbm', {k: (round(v['avg_us_event_timed'], 1), round(v['frac_of_8tb_s'], 3)) for k, v in r['hbm_kernels'].items()})
print('ab', d.get('ab_norm2_fold'))
print('batched', d.get('batched', {}).get('value'), 'parity', d.get('parity_path', {}).get('value'), 'fast', (d.get('parity_path_fast') or {}).get('value'))
print('cpu', d.get('cpu_baseline'), d.get('cpu_baseline_configs0'))
print('other', [(o['config'][:20], round(o['value'], 1)) for o in d.get('other_configs', [])])
PY
timeout 3000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/pytest.log; tail -8 gpurun_out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
