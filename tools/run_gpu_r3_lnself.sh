# LayerNorm statistics inside the consuming GEMMs: unit tests, the parity suites, same-box A/B of the bench
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_lnself.py -m gpu -q --tb=short -p no:cacheprovider -x -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r3_lnself_unit.log; tail -25 gpurun_out/r3_lnself_unit.log
timeout 1500 python -m pytest tests/test_gpu_parity_bf16.py tests/test_gpu_e2e.py tests/test_gpu_plan.py -m gpu -q --tb=short -p no:cacheprovider -x -k "not fp32" 2>&1 | tail -12 > gpurun_out/r3_lnself_parity.log; tail -12 gpurun_out/r3_lnself_parity.log
for rep in 1 2; do
for v in 0 1; do
TOC3D_LN_SELF=$v timeout 600 python bench.py --no-cpu-baseline --no-batched --no-other-configs --no-parity-path > gpurun_out/r3_lnself_bench_$v.json 2> gpurun_out/r3_lnself_bench_$v.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r3_lnself_bench_$v.json').read().strip().splitlines()[-1])
print('LN_SELF=$v', round(d['value'], 2), 'frames/s', round(d['ms_per_step'], 4), 'ms  roofline', round(d['roofline']['frac'], 4), d['roofline'].get('launches_per_step'))
PY
done
done
