# round 2: attention VALU trimming: op tests + timing + e2e bit-stability + bench breakdown
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -k "attention" 2>&1 | tail -5
python tools/attn_pmc.py 20 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_e2e.py -q --tb=short -p no:cacheprovider -k "repeated or vitl or view_groups" 2>&1 | tail -4
rm -f gpurun_out/tune_ship.json
timeout 900 python bench.py --steps 100 --warmup 10 --tune-cache gpurun_out/tune_ship.json --no-cpu-baseline > gpurun_out/r2i_bench.json 2> gpurun_out/r2i_bench.err; head -12 gpurun_out/r2i_bench.err; grep window_attention gpurun_out/r2i_bench.err
python -c "import json;d=json.load(open('gpurun_out/r2i_bench.json'));print(round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms', d['roofline']['frac'])"
