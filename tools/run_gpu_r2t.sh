# round 2, final state: tuned tables, profiles, default bench line, whole GPU suite
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 1800 python tools/make_tuned_tables.py gpurun_out/tuned 2>&1 | grep -v amdgpu
cp gpurun_out/tuned/*.json toc3d_amd/tuned/
bash tools/run_gpu_r2prof.sh r2t 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r2t_bench.json 2> gpurun_out/r2t_bench.err
python -c "import json;d=json.load(open('gpurun_out/r2t_bench.json'));print(round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms', d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('parity_path',{}).get('value'), d.get('cpu_baseline',{}).get('value'))"
timeout 2700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r2t_pytest.log; tail -3 gpurun_out/r2t_pytest.log
