mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -x -k "linear or residual or ffn_ln or norm2 or conv3x3 or stable" 2>&1 | tail -5 > gpurun_out/r3_respf_unit.log; tail -5 gpurun_out/r3_respf_unit.log
for rep in 1 2; do
for v in "0 192" "2 192" "1 192" "4 192" "2 0"; do
set -- $v
TOC3D_RES_PF=$1 TOC3D_PREFETCH=$2 timeout 600 python bench.py --no-cpu-baseline --no-batched --no-other-configs --no-parity-path > gpurun_out/r3_respf_bench.json 2> gpurun_out/r3_respf_bench.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r3_respf_bench.json').read().strip().splitlines()[-1])
print('RES_PF=$1 PREFETCH=$2', round(d['value'], 2), 'frames/s', round(d['ms_per_step'], 4), 'ms', [round(x, 3) for x in d['repetitions']['ms_per_step_each']], 'gemm avg', round(d['roofline']['avg_launch_ms'] * 1e3, 2))
PY
done
done
