# round 2, second GPU pass: whole GPU suite, default bench line, kernel traces of the replayed plan with 1 and 2 view groups
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -30 > gpurun_out/r2b_pytest.log; tail -5 gpurun_out/r2b_pytest.log
timeout 900 python bench.py --tune-cache gpurun_out/tune_r2.json > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err; tail -c 1500 gpurun_out/r2b_bench.json
export TMPDIR=/tmp
for g in 1 2; do
  rm -rf /tmp/kt$g
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$g -o kt -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 4 --groups $g --launch plan \
      --tune-cache "$GRAFT_REPO_ROOT/gpurun_out/tune_r2.json" --no-cpu-baseline --no-breakdown > "$GRAFT_REPO_ROOT/gpurun_out/r2b_trace_g$g.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r2b_trace_g$g.err" )
  f=$(find /tmp/kt$g -name "*kernel_trace.csv" | head -1)
  # keep the last ~3 frames of the trace (the timed steps): small enough to travel back
  if [ -n "$f" ]; then head -1 "$f" > gpurun_out/r2b_kt_g$g.csv; tail -n 2500 "$f" >> gpurun_out/r2b_kt_g$g.csv; fi
  tail -c 300 gpurun_out/r2b_trace_g$g.json
done
ls -la gpurun_out
