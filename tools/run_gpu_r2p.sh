# round 2: conv zero-line fix -- conv / neck tests, tuned tables, profiles, default bench
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -q --tb=short -p no:cacheprovider -k "conv3x3 or neck or loop" -s 2>&1 | grep -E "^\[|passed|failed|Error|assert" | cut -c1-200 | tail -12
timeout 1800 python tools/make_tuned_tables.py gpurun_out/tuned 2>&1 | grep -v amdgpu
cp gpurun_out/tuned/*.json toc3d_amd/tuned/
bash tools/run_gpu_r2prof.sh r2p 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r2p_bench.json 2> gpurun_out/r2p_bench.err; head -20 gpurun_out/r2p_bench.err
python -c "import json;d=json.load(open('gpurun_out/r2p_bench.json'));print(round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms', d['roofline']['frac'], d.get('parity_path',{}).get('value'), d.get('cpu_baseline',{}).get('value'))"
