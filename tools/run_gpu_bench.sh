mkdir -p gpurun_out
python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -s -p no:cacheprovider -k "hires" 2>&1 | tail -6
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-breakdown --hw 640x1600 2>&1 | tail -2 | cut -c1-400
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-breakdown --config eva_dense 2>&1 | tail -2 | cut -c1-400
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-breakdown --config toc3d_fast 2>&1 | tail -2 | cut -c1-400
