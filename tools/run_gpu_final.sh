# final check of HEAD: whole GPU suite, smoke, default bench line (what the driver runs at round end)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 3000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/final_pytest.log; tail -6 gpurun_out/final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
timeout 1200 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/final_bench.json').read().strip().splitlines()[-1])
print(round(d['value'], 1), 'frames/s', round(d['ms_per_step'], 3), 'ms  roofline', round(d['roofline']['frac'], 4), round(d['roofline'].get('frac_issued', 0), 4), 'batched', d.get('batched', {}).get('value'),
      'parity', d.get('parity_path', {}).get('value'), 'fast', (d.get('parity_path_fast') or {}).get('value'), 'cpu', d.get('cpu_baseline', {}).get('value'))
print('other', [(o['config'][:20], round(o['value'], 1)) for o in d.get('other_configs', [])])
PY
# kernel trace of the same bench command (profiles/r03_kernel_stats.csv is summarised from it by tools/summarize_prof.py)
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
rm -rf $OUT/kt
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs > $OUT/final_prof_bench.json 2> $OUT/final_prof.err
for f in $(find $OUT/kt -mindepth 2 -name "kt_*.csv"); do cp $f $OUT/kt/; done
find $OUT/kt -mindepth 1 -type d -exec rm -rf {} + 2>/dev/null
ls $OUT/kt | head
