# round 6, first GPU pass: the x3 pre-rotated attention tests, the fp32x3 e2e goldens, then the fp32x3 bench leg
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_attn_rot.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -25 | tee gpurun_out/r6_attn_rot.txt
timeout 1500 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -k "fp32x3 or vitl_fp32 or x3" 2>&1 | tail -25 | tee gpurun_out/r6_e2e_x3.txt
timeout 600 python bench.py --precision fp32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-parity-path --no-other-configs --no-ab 2>gpurun_out/r6_bench_x3.err | tail -1 > gpurun_out/r6_bench_x3.json
python -c "
import json; d=json.loads(open('gpurun_out/r6_bench_x3.json').read()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"
