# one library, one environment variable, several values: N alternations of the default bench step (VAR=name VALS="a b c" [BENCH_ARGS=...] bash tools/gpu/env_ab.sh)
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
N=${N:-3}
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs --no-ab --no-calibration $BENCH_ARGS"
for i in $(seq 1 $N); do
  for V in $VALS; do
    env $VAR=$V $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$V', round(d['value'],2), [round(x,4) for x in d['repetitions']['ms_per_step_each']])"
  done
done | tee gpurun_out/env_ab.txt
python - <<'PY'
v={}
for l in open('gpurun_out/env_ab.txt'):
    k,x=l.split()[:2]; v.setdefault(k,[]).append(float(x))
for k,x in v.items(): print(k, 'median', sorted(x)[len(x)//2], 'mean', round(sum(x)/len(x),2))
PY
