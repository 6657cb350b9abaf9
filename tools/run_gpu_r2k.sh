# round 2: norm2 fold -- op tests, e2e + bf16 parity, same-box A/B (norm2 fold off / on)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -x -k "folded or bit_identical or bit_stable" -s 2>&1 | tail -12 > gpurun_out/r2k_ops.log; cat gpurun_out/r2k_ops.log
for c in n2off n2on; do cp toc3d_amd/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tune_$c.json; done
run() { TOC3D_FOLD_N2=$2 timeout 600 python bench.py --steps 100 --warmup 10 --tune-cache gpurun_out/tune_$3.json --no-cpu-baseline --no-breakdown --no-parity-path > gpurun_out/r2k_$1.json 2> gpurun_out/r2k_$1.err
  python -c "import json;d=json.load(open('gpurun_out/r2k_$1.json'));print('$1', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms')"; }
for rep in 1 2 3; do run n2off_$rep 0 n2off; run n2on_$rep 1 n2on; done
timeout 2400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_bf16.py tests/test_gpu_plan.py -q --tb=short -p no:cacheprovider -x -s 2>&1 | grep -E "^\[|passed|failed|Error|assert" | cut -c1-260 | tail -30 > gpurun_out/r2k_e2e.log; cat gpurun_out/r2k_e2e.log
