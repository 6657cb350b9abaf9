# round 3: pre-rotated attention -- new op tests first, then the suites that exercise the changed paths, then a bench with the per-op breakdown
mkdir -p gpurun_out
python -m pytest tests/test_gpu_attn_rot.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -25 > gpurun_out/r3b_rot.log; tail -25 gpurun_out/r3b_rot.log
python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_gpu_parity_bf16.py tests/test_gpu_plan.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -25 > gpurun_out/r3b_tests.log; tail -8 gpurun_out/r3b_tests.log
python bench.py --no-cpu-baseline --no-batched --no-parity-path > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err; cut -c1-300 gpurun_out/r3b_bench.json; head -24 gpurun_out/r3b_bench.err | tail -22
TOC3D_ATTN_ROT=0 python bench.py --no-cpu-baseline --no-batched --no-parity-path --no-breakdown > gpurun_out/r3b_bench_norot.json 2> /dev/null; cut -c1-200 gpurun_out/r3b_bench_norot.json
