# final check of HEAD: whole GPU suite, smoke, default bench line
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 2700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/final_pytest.log; tail -3 gpurun_out/final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
python -c "import json;d=json.load(open('gpurun_out/final_bench.json'));print(round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms', round(d['roofline']['frac'],4), d['roofline']['avg_launch_ms'], d.get('batched',{}).get('value'), d.get('parity_path',{}).get('value'), d.get('cpu_baseline',{}).get('value'))"
