mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -120 > gpurun_out/ops.log
python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -s -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/e2e.log
tail -5 gpurun_out/ops.log; tail -5 gpurun_out/e2e.log
