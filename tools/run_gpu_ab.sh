# same-box A/B of builds: each argument is a directory holding a full tree with its built library
for rep in 1 2; do
for d in "$@"; do
  echo "== $d"
  (cd $d && python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/tmp/ab.err | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('  frames/s', round(d['value'], 1), ' gemm TF', round(d['roofline']['achieved']), ' gemm ms', round(d['roofline']['avg_launch_ms'] * d['roofline']['launches_per_step'], 3))"; grep -E "epi2 .*M=6000|epi2 .*M=3744" /tmp/ab.err | cut -c1-150)
done
done
