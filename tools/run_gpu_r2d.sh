# round 2: ffn_ln fold -- op test, e2e parity, same-box A/B of the bench (fold off / on, three repetitions each)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -x -k "ffn_ln or copy_segments or variants_are_bit" -s 2>&1 | tail -15 > gpurun_out/r2d_ops.log; cat gpurun_out/r2d_ops.log
cp toc3d_amd/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tune_fold.json
cp toc3d_amd/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tune_nofold.json
for rep in 1 2 3; do
  for f in 0 1; do
    tc=gpurun_out/tune_fold.json; [ $f = 0 ] && tc=gpurun_out/tune_nofold.json
    TOC3D_FOLD_LN=$f timeout 600 python bench.py --steps 100 --warmup 10 --tune-cache $tc --no-cpu-baseline --no-breakdown > gpurun_out/r2d_bench_f${f}_$rep.json 2> gpurun_out/r2d_bench_f${f}_$rep.err
    python -c "import json;d=json.load(open('gpurun_out/r2d_bench_f${f}_$rep.json'));print('fold=$f rep $rep', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms')"
  done
done
timeout 1800 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_bf16.py tests/test_gpu_plan.py -q --tb=short -p no:cacheprovider -x -s 2>&1 | grep -E "^\[|passed|failed|Error|assert" | cut -c1-300 | tail -40 > gpurun_out/r2d_e2e.log; cat gpurun_out/r2d_e2e.log
