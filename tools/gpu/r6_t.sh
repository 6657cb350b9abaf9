export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
T0=$(date +%s); timeout 1500 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "default bench.py wall time: $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if isinstance(v,(int,float))})
PY
