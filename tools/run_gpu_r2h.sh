# round 2: variant 51 on the frame's shapes; profiles (kernel stats, traffic, SQ counters of GEMM + attention)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
VARIANTS=16,116,51,151,17,117,45,145,49,149,126,114 timeout 900 python tools/gemm_phased_check.py time 2>&1 | grep -v amdgpu > gpurun_out/r2h_v51.log; cat gpurun_out/r2h_v51.log
bash tools/run_gpu_r2prof.sh r2h 2>&1 | tail -15
