# round 3 baseline: GPU test suite + default bench on a fresh box (what the driver will run)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/r3base_tests.log; tail -4 gpurun_out/r3base_tests.log
python bench.py > gpurun_out/r3base_bench.json 2> gpurun_out/r3base_bench.err; cut -c1-600 gpurun_out/r3base_bench.json
