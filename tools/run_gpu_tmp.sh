cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_bf16.py -m gpu -q --tb=short -p no:cacheprovider -k "fp32x3" 2>&1 | tail -30
