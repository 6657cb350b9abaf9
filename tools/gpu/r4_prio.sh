mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
python tools/ubench/schedule_ab.py "env:TOC3D_SIDE_PRIO=low" "env:TOC3D_SIDE_PRIO=default" bf16 toc3d_faster 320x800 5 40 2>&1 | grep "^#" | tee gpurun_out/r4_prio_ab.txt
python tools/ubench/schedule_ab.py "env:TOC3D_SIDE_PRIO=high" "env:TOC3D_SIDE_PRIO=default" bf16 toc3d_faster 320x800 5 40 2>&1 | grep "^#" | tee -a gpurun_out/r4_prio_ab.txt
