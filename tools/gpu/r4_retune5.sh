# round 4: in-place passes on the exact-f32 table (first) and the fp32x3 table (second)
mkdir -p gpurun_out/tuned
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 2400 python tools/tune_in_place.py toc3d_amd/tuned/toc3d_faster_320x800_fp32.json gpurun_out/tuned/toc3d_faster_320x800_fp32_inplace.json toc3d_faster 8 tiles 320x800 fp32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_retune5_fp32.txt
timeout 2400 python tools/tune_in_place.py toc3d_amd/tuned/toc3d_faster_320x800_fp32x3.json gpurun_out/tuned/toc3d_faster_320x800_fp32x3_pass2.json toc3d_faster 16 tiles 320x800 fp32x3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_retune5_x3.txt
