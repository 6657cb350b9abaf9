# generic: TESTS="..." [K="-k expr"] [OUT=name] bash tools/gpu/r6_run.sh  -- runs the given pytest selection on the GPU box and keeps the tail under gpurun_out/
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout ${TMO:-3000} python -m pytest $TESTS -m gpu -q --tb=short -p no:cacheprovider ${K:+-k "$K"} 2>&1 | grep -v "amdgpu.ids" | tail -${TAIL:-30} | tee gpurun_out/${OUT:-r6_run}.txt
