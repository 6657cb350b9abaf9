mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_attn_rot.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -12 > gpurun_out/r3h_tests.log; tail -4 gpurun_out/r3h_tests.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-batched --no-parity-path --no-breakdown --no-other-configs 2>/dev/null | cut -c1-220
TOC3D_WIDE_STORES=0 python bench.py --no-cpu-baseline --no-batched --no-parity-path --no-breakdown --no-other-configs 2>/dev/null | cut -c1-220
done
