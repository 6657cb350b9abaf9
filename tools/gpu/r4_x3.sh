mkdir -p gpurun_out/tuned
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -k "bf16x3" -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -5
TOC3D_KEEP_TABLE=1 timeout 2400 python tools/make_tuned_tables.py gpurun_out/tuned "toc3d_faster:320x800:fp32x3" "toc3d_fast:320x800:fp32x3" "eva_dense:320x800:fp32x3" "toc3d_faster:640x1600:fp32x3" 2>&1 | grep -v amdgpu.ids | tail -5
cp gpurun_out/tuned/*fp32x3.json toc3d_amd/tuned/
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_bf16.py -k "fp32x3" -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8
python tools/ubench/schedule_ab.py "fold_norm2=True,carry_compact=True" "fold_norm2=False,carry_compact=False" fp32x3 toc3d_faster 320x800 3 10 2>&1 | grep "^#" | tee gpurun_out/r4_x3_ab.txt
python tools/ubench/schedule_ab.py "prefetch_weights=192" "prefetch_weights=0" fp32x3 toc3d_faster 320x800 3 10 2>&1 | grep "^#" | tee -a gpurun_out/r4_x3_ab.txt
