# generic: TESTS="..." [K="-k expr"] bash tools/gpu/r6_run.sh  -- runs the given pytest selection and tails the output
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout ${TMO:-2400} python -m pytest $TESTS -m gpu -q --tb=short -p no:cacheprovider -s ${K:+-k "$K"} 2>&1 | grep -v "amdgpu.ids" | tail -${TAIL:-60} | tee gpurun_out/${OUT:-r6_run}.txt
