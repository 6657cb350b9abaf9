# development timelines cited by the kernel sources (regenerated on a fresh box) + the A/B of the cross-frame weight prefetch
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 300 tools/ubench/bin/gemm_timeline 0 > gpurun_out/r3_gemm_timeline_warm.txt 2>&1; tail -3 gpurun_out/r3_gemm_timeline_warm.txt
timeout 300 tools/ubench/bin/gemm_timeline 1 > gpurun_out/r3_gemm_timeline_cold.txt 2>&1; tail -3 gpurun_out/r3_gemm_timeline_cold.txt
timeout 300 tools/ubench/bin/gemm_timeline 0 1 > gpurun_out/r3_gemm_timeline_rope.txt 2>&1; tail -3 gpurun_out/r3_gemm_timeline_rope.txt
timeout 300 tools/ubench/bin/cu_split > gpurun_out/r3_cu_split.txt 2>&1; tail -5 gpurun_out/r3_cu_split.txt
timeout 600 python tools/ubench/attn_timeline.py > gpurun_out/r3_attn_timeline.txt 2>&1; tail -5 gpurun_out/r3_attn_timeline.txt
for rep in 1 2; do
for v in 0 1; do
TOC3D_PREFETCH_WRAP=$v timeout 600 python bench.py --no-cpu-baseline --no-batched --no-other-configs --no-parity-path > gpurun_out/r3_wrap_bench_$v.json 2> gpurun_out/r3_wrap_bench_$v.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r3_wrap_bench_$v.json').read().strip().splitlines()[-1])
print('PREFETCH_WRAP=$v', round(d['value'], 2), 'frames/s', round(d['ms_per_step'], 4), 'ms', d['repetitions']['ms_per_step_each'])
PY
done
done
