# round 2: tuned tables for the BASELINE configs, then the profiles and the default bench line with them
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_e2e.py -q --tb=short -p no:cacheprovider -k "neck or loop" 2>&1 | tail -3
timeout 1800 python tools/make_tuned_tables.py gpurun_out/tuned 2>&1 | grep -v amdgpu
cp gpurun_out/tuned/*.json toc3d_amd/tuned/
bash tools/run_gpu_r2prof.sh r2j 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/r2j_bench.json 2> gpurun_out/r2j_bench.err; head -14 gpurun_out/r2j_bench.err
python -c "import json;d=json.load(open('gpurun_out/r2j_bench.json'));print(round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms', d['roofline']['frac'], d.get('parity_path'))"
