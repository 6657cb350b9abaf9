# same-box A/B of three builds: pre-chain (7da5dd1), chain with the first epilogue (821cd75), current tree
for rep in 1 2; do
for d in _ab/pre_chain _ab/chain_fat .; do
  echo "== $d"
  (cd $d && python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/tmp/ab.err | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('  frames/s', round(d['value'], 1), ' gemm TF', round(d['roofline']['achieved']), ' gemm ms', round(d['roofline']['avg_launch_ms'] * d['roofline']['launches_per_step'], 3))"; grep -E "window_attention  |layernorm_rows  " /tmp/ab.err | cut -c1-80)
done
done
