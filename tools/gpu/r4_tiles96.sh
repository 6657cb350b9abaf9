export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_ops.py -k "variants_are_bit_identical or folded or bit_stable_under" tests/test_gpu_attn_rot.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -6
timeout 1500 python tools/ubench/variant_ab.py 54,55,56,57,58,154,155,156,157,158 toc3d_faster 320x800 5 > gpurun_out/r4_tiles96_ab.txt 2> gpurun_out/r4_tiles96_ab.err; cat gpurun_out/r4_tiles96_ab.txt; tail -3 gpurun_out/r4_tiles96_ab.err
