# re-measure the shipped bf16 tile tables with the 2-D XCD orders among the candidates (variants 2xx / 3xx), refine the headline table in place, A/B against the old table
mkdir -p gpurun_out/tuned
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -x -k "variants_are_bit_identical or ffn_ln or norm2" 2>&1 | tail -3
timeout 1500 python tools/make_tuned_tables.py gpurun_out/tuned toc3d_faster:320x800:bf16:1,2 toc3d_fast:320x800:bf16 eva_dense:320x800:bf16 toc3d_faster:640x1600:bf16 2>&1 | grep -v amdgpu.ids | tail -6
cp gpurun_out/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tuned/toc3d_faster_320x800_bf16_cold.json
timeout 1800 python tools/tune_in_place.py gpurun_out/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tuned/toc3d_faster_320x800_bf16.json toc3d_faster 40 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r3_tune_in_place.log; tail -25 gpurun_out/r3_tune_in_place.log
for rep in 1 2; do
for t in shipped cold inplace; do
case $t in
 shipped) TC="";;
 cold) cp gpurun_out/tuned/toc3d_faster_320x800_bf16_cold.json /tmp/t.json; TC="--tune-cache /tmp/t.json";;
 inplace) cp gpurun_out/tuned/toc3d_faster_320x800_bf16.json /tmp/t.json; TC="--tune-cache /tmp/t.json";;
esac
timeout 600 python bench.py $TC --no-cpu-baseline --no-batched --no-other-configs --no-parity-path > gpurun_out/r3_retune_bench.json 2> gpurun_out/r3_retune_bench.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r3_retune_bench.json').read().strip().splitlines()[-1])
print('table=$t', round(d['value'], 2), 'frames/s', [round(x, 3) for x in d['repetitions']['ms_per_step_each']], 'gemm avg us', round(d['roofline']['avg_launch_ms'] * 1e3, 2), 'frac', round(d['roofline']['frac'], 4))
PY
done
done
