# round 4, run 1: the 32x32x16-MFMA tile variants -- parity tests, then cold per-launch times and an alternating frame-time A/B against the shipped table
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_mfma32.py tests/test_gpu_plan.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -25 > gpurun_out/r4_mfma32_pytest.log; tail -8 gpurun_out/r4_mfma32_pytest.log
timeout 1500 python tools/ubench/variant_ab.py 70,71,72,73,74,75,76,77,78,79,80,81,82,83 toc3d_faster 320x800 5 > gpurun_out/r4_mfma32_ab.txt 2> gpurun_out/r4_mfma32_ab.err; tail -45 gpurun_out/r4_mfma32_ab.txt; tail -5 gpurun_out/r4_mfma32_ab.err
