# memory-side traffic of the GEMM launches of the shipped default (norm2 fold on): FETCH_SIZE and WRITE_SIZE in two separate --pmc passes
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/d2"
mkdir -p $OUT
CMD="python bench.py --steps 4 --warmup 2 --reps 1 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs"
for pass in "fs FETCH_SIZE" "wsz WRITE_SIZE"; do
  set -- $pass; d=$1; shift
  rm -rf $OUT/$d
  timeout 100 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$d -o $d --output-format csv -- $CMD > $OUT/pmc_$d.json 2> $OUT/pmc_$d.err
  for f in $(find $OUT/$d -mindepth 2 -name "${d}_*.csv"); do cp $f $OUT/$d/; done
  find $OUT/$d -mindepth 1 -type d -exec rm -rf {} + 2>/dev/null
  find $OUT/$d -name "*kernel_trace.csv" -delete
  ls -la $OUT/$d | head -5
done
