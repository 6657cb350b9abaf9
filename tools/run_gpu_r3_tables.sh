# tile tables for the (config, size, precision) points that had none: every shape of a first frame, 1600x800, toc3d_fast fp32, dense fp32 / x3
mkdir -p gpurun_out/tuned2
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 2400 python tools/make_tuned_tables.py gpurun_out/tuned2 toc3d_faster:800x1600:bf16 toc3d_fast:320x800:fp32 eva_dense:320x800:fp32 toc3d_fast:320x800:fp32x3 eva_dense:320x800:fp32x3 toc3d_faster:640x1600:fp32x3 2>&1 | grep -v amdgpu.ids | tail -8
