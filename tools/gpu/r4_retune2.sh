mkdir -p gpurun_out/tuned
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 2400 python tools/tune_in_place.py toc3d_amd/tuned/toc3d_faster_640x1600_bf16.json gpurun_out/tuned/toc3d_faster_640x1600_bf16_inplace.json toc3d_faster 12 tiles 640x1600 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_retune_1600.txt
timeout 1800 python tools/tune_in_place.py toc3d_amd/tuned/eva_dense_320x800_bf16.json gpurun_out/tuned/eva_dense_320x800_bf16_inplace.json eva_dense 40 tiles 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_retune_dense.txt
