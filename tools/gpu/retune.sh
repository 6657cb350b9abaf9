# In-place tuning pass of one tile table on the GPU box (tools/tune_in_place.py): frame-time greedy, a change must win twice.
#   gpurun -- 'bash tools/gpu/retune.sh <table under toc3d_amd/tuned> <config> <frames per measurement> <tiles|orders> <HxW> [precision] [frames per forward]'
# The refined table lands in gpurun_out/tuned/<table>; copy it over the shipped one after an A/B (tools/gpu/table_ab.sh).
mkdir -p gpurun_out/tuned
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
T=$1; shift
timeout 2400 python tools/tune_in_place.py toc3d_amd/tuned/$T gpurun_out/tuned/$T "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/retune_${T%.json}.txt
