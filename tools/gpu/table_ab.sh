# A/B of two tile tables by frame time (N alternations of the default bench step, separate processes): tools/gpu/base_table.json vs tools/gpu/cand_table.json
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs --no-ab --no-calibration $BENCH_ARGS"
for i in $(seq 1 ${N:-5}); do
  for T in base cand; do
    cp tools/gpu/${T}_table.json /tmp/t_$T.json
    $B --tune-cache /tmp/t_$T.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$T', round(d['value'],2), [round(x,4) for x in d['repetitions']['ms_per_step_each']])"
  done
done | tee gpurun_out/table_ab.txt
python - <<'PY'
v={}
for l in open('gpurun_out/table_ab.txt'):
    k,x=l.split()[:2]; v.setdefault(k,[]).append(float(x))
for k,x in v.items(): print(k, 'median', sorted(x)[len(x)//2], 'mean', round(sum(x)/len(x),2))
PY
