mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -k "gather_merge" -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
for sp in 2 8 16; do timeout 900 python tools/ubench/schedule_ab.py "gather_split=$sp" "gather_split=True" bf16 toc3d_faster 320x800 5 40 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_gather_ab.txt; done
