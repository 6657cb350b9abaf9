mkdir -p gpurun_out
python -m pytest tests/test_gpu_plan.py tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gumbel or bf16x3 or plan" -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r3i_a.log; tail -14 gpurun_out/r3i_a.log
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_bf16.py -m gpu -q --tb=short -p no:cacheprovider -x -k "fp32x3 or carried or first_frame or forced" -s 2>&1 | grep -E "rel max|passed|failed|Error|error" | tail -30 > gpurun_out/r3i_b.log; tail -30 gpurun_out/r3i_b.log
python bench.py --no-cpu-baseline --no-batched --no-other-configs > gpurun_out/r3i_bench.json 2> gpurun_out/r3i_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3i_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d.get('parity_path'), d.get('parity_path_fast'))
print(d['roofline']['frac'], d['roofline'].get('frac_issued'), d['roofline'].get('hbm_kernels'))
PY
