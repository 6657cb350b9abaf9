# round 2: ffn_ln fold with the cheaper epilogues + weight prefetch lane: tests, then same-box A/B (two repetitions each)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -x -k "ffn_ln or copy_segments or variants_are_bit or bit_stable" -s 2>&1 | tail -8 > gpurun_out/r2f_ops.log; cat gpurun_out/r2f_ops.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q --tb=short -p no:cacheprovider -x -k "repeated or view_groups or vitl" 2>&1 | tail -5 > gpurun_out/r2f_e2e.log; cat gpurun_out/r2f_e2e.log
for c in nofold fold; do cp toc3d_amd/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tune_$c.json; done
run() { # tag fold prefetch tunefile
  TOC3D_FOLD_LN=$2 TOC3D_PREFETCH=$3 timeout 600 python bench.py --steps 100 --warmup 10 --tune-cache gpurun_out/tune_$4.json --no-cpu-baseline --no-breakdown > gpurun_out/r2f_$1.json 2> gpurun_out/r2f_$1.err
  python -c "import json;d=json.load(open('gpurun_out/r2f_$1.json'));print('$1', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms')"
}
for rep in 1 2; do
  run nofold_$rep 0 0 nofold
  run fold_$rep 1 0 fold
  run fold_pf8_$rep 1 8 fold
  run fold_pf32_$rep 1 32 fold
  run fold_pf128_$rep 1 128 fold
done
timeout 900 python bench.py --tune-cache gpurun_out/tune_fold.json --no-cpu-baseline > gpurun_out/r2f_bench_breakdown.json 2> gpurun_out/r2f_bench_breakdown.err; head -24 gpurun_out/r2f_bench_breakdown.err
