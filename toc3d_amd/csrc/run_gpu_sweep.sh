mkdir -p gpurun_out
python tools/gemm_sweep.py > gpurun_out/gemm_sweep5.txt 2>&1; cat gpurun_out/gemm_sweep5.txt
