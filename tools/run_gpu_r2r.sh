# round 2: after splitting the GEMM instantiations over translation units -- whole GPU suite, smoke, default bench
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 2700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r2r_pytest.log; tail -4 gpurun_out/r2r_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
for rep in 1 2; do
timeout 900 python bench.py --no-cpu-baseline --no-parity-path > gpurun_out/r2r_bench_$rep.json 2> gpurun_out/r2r_bench_$rep.err
python -c "import json;d=json.load(open('gpurun_out/r2r_bench_$rep.json'));print(round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms', round(d['roofline']['frac'],4))"
done
