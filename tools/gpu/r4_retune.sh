mkdir -p gpurun_out/tuned
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 3000 python tools/tune_in_place.py toc3d_amd/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tuned/toc3d_faster_320x800_bf16_inplace.json toc3d_faster 40 tiles 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_retune_800.txt
