mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/r4_gather_ab.txt
timeout 900 python tools/ubench/schedule_ab.py "gather_split=True" "gather_split=True" bf16 toc3d_faster 320x800 5 40 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_gather_ab.txt
for sp in False 2 8; do timeout 900 python tools/ubench/schedule_ab.py "gather_split=True" "gather_split=$sp" bf16 toc3d_faster 320x800 5 40 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_gather_ab.txt; done
