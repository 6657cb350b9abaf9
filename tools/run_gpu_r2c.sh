# round 2: skip-and-measure (what each kernel family / cold weights cost the frame in place)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 1500 python tools/ubench/where_time_goes.py gpurun_out/tune_r2.json > gpurun_out/r2c_where.log 2>&1; cat gpurun_out/r2c_where.log | tail -30
