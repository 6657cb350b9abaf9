mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/tests.log; tail -6 gpurun_out/tests.log
python bench.py --steps 50 --warmup 10 > gpurun_out/bench.json 2> gpurun_out/bench.err; grep -v "linear_ex\[" gpurun_out/bench.err | head -22 | cut -c1-150; cut -c1-1700 gpurun_out/bench.json
