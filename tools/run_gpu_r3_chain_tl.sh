mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
{ python tools/ubench/chain_timeline.py 2898 0 8 768 0 1; python tools/ubench/chain_timeline.py 2898 2 8 512 0 1; python tools/ubench/chain_timeline.py 2898 0 8 256 0 1;  python tools/ubench/chain_timeline.py 6000 2 8 512 0 2; } > gpurun_out/r3_chain_tl.log 2>&1; cat gpurun_out/r3_chain_tl.log
