# round 2: whole GPU suite after the ffn_ln fold fix; phased-GEMM ceiling microbenchmark; bench with tuning saved
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 600 python tools/ubench/phased_ceiling.py > gpurun_out/r2e_phased_ceiling.log 2>&1; cat gpurun_out/r2e_phased_ceiling.log
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/r2e_pytest.log; tail -12 gpurun_out/r2e_pytest.log
cp toc3d_amd/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tune_fold.json
timeout 900 python bench.py --tune-cache gpurun_out/tune_fold.json > gpurun_out/r2e_bench.json 2> gpurun_out/r2e_bench.err; tail -c 1200 gpurun_out/r2e_bench.json; head -30 gpurun_out/r2e_bench.err
