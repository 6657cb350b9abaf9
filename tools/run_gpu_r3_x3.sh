mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -x -k "x3 or x6 or conv3x3 or linear_bias" -s 2>&1 | grep -E "passed|failed|bf16x3|Error|assert" | tail -8
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_bf16.py -m gpu -q --tb=short -p no:cacheprovider -x -k "fp32x3" -s 2>&1 | grep -E "rel max|passed|failed|Error|error" | tail -12
for p in fp32x3 fp32x6; do
timeout 600 python bench.py --precision $p --steps 10 --warmup 3 --no-cpu-baseline --no-batched --no-other-configs --no-parity-path --no-breakdown > gpurun_out/r3_x3_bench_$p.json 2> gpurun_out/r3_x3_bench_$p.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r3_x3_bench_$p.json').read().strip().splitlines()[-1])
print('$p', round(d['value'], 2), 'frames/s', [round(x, 3) for x in d['repetitions']['ms_per_step_each']])
PY
done
