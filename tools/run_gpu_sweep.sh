mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "variants" 2>&1 | tail -3
python tools/gemm_sweep.py > gpurun_out/gemm_sweep4.txt 2>&1; cat gpurun_out/gemm_sweep4.txt
