# round 4, third in-place pass on the 800x320 table (after the batched statistics loads changed the LN-consuming launches' costs), then the orders pass
mkdir -p gpurun_out/tuned
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 1800 python tools/tune_in_place.py toc3d_amd/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tuned/toc3d_faster_320x800_bf16_pass3.json toc3d_faster 40 tiles 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_retune3_800.txt
timeout 1200 python tools/tune_in_place.py gpurun_out/tuned/toc3d_faster_320x800_bf16_pass3.json gpurun_out/tuned/toc3d_faster_320x800_bf16_pass3o.json toc3d_faster 40 orders 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_retune3_800_orders.txt
