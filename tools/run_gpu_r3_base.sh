# round 3 baseline of HEAD on a fresh box: whole GPU suite, smoke, the default bench line, and the rocprofv3 kernel trace of the bench command
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 2700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/r3_pytest.log; tail -4 gpurun_out/r3_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee gpurun_out/r3_smoke.log
timeout 1200 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, 'parity', d.get('parity_path', {}).get('value'), 'fast', d.get('parity_path_fast'), 'batched', d.get('batched', {}).get('value'))
print('roofline', d['roofline']['frac'], d['roofline'].get('frac_issued'), d['roofline'].get('avg_launch_ms'))
print('hbm_kernels', d['roofline'].get('hbm_kernels'))
print('other', d.get('other_configs'))
PY
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
rm -rf $OUT/kt
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs > $OUT/r3_prof_bench.json 2> $OUT/r3_prof.err
find $OUT/kt -name "*kernel_stats.csv" | head; find $OUT/kt -name "*kernel_trace.csv" -size +60M -delete
for f in $(find $OUT/kt -mindepth 2 -name "kt_*.csv"); do cp $f $OUT/kt/; done; ls -la $OUT/kt | head
