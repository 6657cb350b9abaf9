# round 4: in-place passes on the fp32x3 table (first) and the 1600x640 bf16 table (third)
mkdir -p gpurun_out/tuned
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 2400 python tools/tune_in_place.py toc3d_amd/tuned/toc3d_faster_320x800_fp32x3.json gpurun_out/tuned/toc3d_faster_320x800_fp32x3_inplace.json toc3d_faster 16 tiles 320x800 fp32x3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_retune4_x3.txt
timeout 2400 python tools/tune_in_place.py toc3d_amd/tuned/toc3d_faster_640x1600_bf16.json gpurun_out/tuned/toc3d_faster_640x1600_bf16_pass3.json toc3d_faster 12 tiles 640x1600 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_retune4_1600.txt
