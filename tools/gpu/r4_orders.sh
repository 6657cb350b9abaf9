mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 2400 python tools/tune_in_place.py toc3d_amd/tuned/toc3d_faster_640x1600_bf16.json gpurun_out/tuned/toc3d_faster_640x1600_bf16.json toc3d_faster 12 orders 640x1600 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_orders_1600.txt
