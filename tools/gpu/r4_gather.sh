mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -k "gather_merge" tests/test_gpu_e2e.py -k "gather_merge or tiny_toc3d_fp32_matches or repeated_forwards or carried" -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -8
timeout 900 python tools/ubench/schedule_ab.py "gather_split=True" "gather_split=False" bf16 toc3d_faster 320x800 5 40 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_gather_ab.txt
