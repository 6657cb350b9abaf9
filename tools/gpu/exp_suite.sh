# the WHOLE GPU suite on the EXPERIMENTAL=1 library (chains, K-loop LayerNorm statistics, pending-scatter gather, 32x32 MFMA form included)
export PYTHONUNBUFFERED=1 TOC3D_LIB=libtoc3d_gfx950_exp.so
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -12 | tee gpurun_out/exp_suite.txt
