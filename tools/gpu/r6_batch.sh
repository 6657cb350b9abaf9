export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tuned
timeout 1800 python -m pytest tests/test_gpu_parity_bf16.py -m gpu -q --tb=short -p no:cacheprovider -s -k "stress or 1600_fp32" 2>&1 | grep -v "amdgpu.ids" | tail -40 | tee gpurun_out/r6_parity_new.txt
T=toc3d_faster_320x800_fp32x3.json
TOC3D_TUNE_CANDS=16,116,17,117,49,149,1,45,145,47,147,52,152,28,29,19,22,26,14,114,10,9,33,53 timeout 2400 python tools/tune_in_place.py toc3d_amd/tuned/$T gpurun_out/tuned/$T toc3d_faster 30 tiles 320x800 fp32x3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_retune_x3_all.txt | tail -30
