# round 2: attention with in-register P (S^T = K Q^T): op tests, e2e parity, bench with breakdown
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -k "attention" -s 2>&1 | tail -25 > gpurun_out/r2g_ops.log; cat gpurun_out/r2g_ops.log
python tools/attn_pmc.py 20 2>&1 | tail -8
timeout 1800 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_bf16.py -q --tb=short -p no:cacheprovider -s 2>&1 | grep -E "^\[|passed|failed|Error|assert" | cut -c1-260 | tail -40 > gpurun_out/r2g_e2e.log; cat gpurun_out/r2g_e2e.log
cp toc3d_amd/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tune_fold.json
timeout 900 python bench.py --steps 100 --warmup 10 --tune-cache gpurun_out/tune_fold.json --no-cpu-baseline > gpurun_out/r2g_bench.json 2> gpurun_out/r2g_bench.err; head -12 gpurun_out/r2g_bench.err; grep window_attention gpurun_out/r2g_bench.err
python -c "import json;d=json.load(open('gpurun_out/r2g_bench.json'));print(round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms')"
