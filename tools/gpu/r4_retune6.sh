# round 4: in-place pass on the 800x320 bf16 table with TWO frames per forward (the `batched` leg of bench.py; its shapes came from the cold-launch autotuner only)
mkdir -p gpurun_out/tuned
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 2400 python tools/tune_in_place.py toc3d_amd/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tuned/toc3d_faster_320x800_bf16_b2.json toc3d_faster 20 tiles 320x800 bf16 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_retune6_b2.txt
