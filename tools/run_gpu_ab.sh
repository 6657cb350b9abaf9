# same-box A/B: each argument is "dir[:ENV=VAL]" -- a full tree with its built library, optionally one environment override
for rep in 1 2 3; do
for spec in "$@"; do
  d=${spec%%:*}; e=""; [ "$spec" != "$d" ] && e=${spec#*:}
  echo "== $spec"
  (cd $d && env $e python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/tmp/ab.err | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('  frames/s', round(d['value'], 1), ' gemm TF', round(d['roofline']['achieved']), ' gemm ms', round(d['roofline']['avg_launch_ms'] * d['roofline']['launches_per_step'], 3))")
done
done
