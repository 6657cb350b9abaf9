# chain kernel: correctness (bit-identity with the separate launches) + timing report
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_chain.py -m gpu -q --tb=short -p no:cacheprovider -x -s -k "${1:-chain}" 2>&1 | grep -v "^$" | tail -80 > gpurun_out/r3_chain.log; tail -70 gpurun_out/r3_chain.log
