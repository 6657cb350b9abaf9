# library A/B by frame time: libtoc3d_prev.so (built from HEAD) against libtoc3d_gfx950.so (working tree), N alternations of the default bench step
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
N=${N:-5}
TESTS=${TESTS:-}
if [ -n "$TESTS" ]; then timeout 1200 python -m pytest $TESTS -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4; fi
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs --no-ab --no-calibration $BENCH_ARGS"
for i in $(seq 1 $N); do
  for L in libtoc3d_prev.so libtoc3d_gfx950.so; do
    TOC3D_LIB=$L $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value'],2), [round(x,4) for x in d['repetitions']['ms_per_step_each']])"
  done
done | tee gpurun_out/lib_ab.txt
python - <<'PY'
import re
v={}
for l in open('gpurun_out/lib_ab.txt'):
    k,x=l.split()[:2]; v.setdefault(k,[]).append(float(x))
for k,x in v.items(): print(k, 'median', sorted(x)[len(x)//2], 'mean', sum(x)/len(x))
PY
