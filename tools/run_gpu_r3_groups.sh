# two view groups under a recorded plan were much slower than eager launches in round 2 (127 vs 186 frames/s): 2 G + 1 = 5 lanes = 5 HIP streams on the
# runtime's default 4 hardware queues?  A/B of GPU_MAX_HW_QUEUES for --groups 2 (plan / eager) and for the shipped --groups 1.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
for cfg in "1 plan 4" "1 plan 8" "2 plan 4" "2 plan 8" "2 eager 4" "2 eager 8" "2 graph 4" "2 graph 8" "3 plan 8"; do
set -- $cfg
GPU_MAX_HW_QUEUES=$3 timeout 600 python bench.py --groups $1 --launch $2 --steps 30 --warmup 5 --no-cpu-baseline --no-batched --no-other-configs --no-parity-path --no-breakdown > gpurun_out/r3_groups_bench.json 2> gpurun_out/r3_groups_bench.err
python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r3_groups_bench.json').read().strip().splitlines()[-1])
    print('groups=$1 launch=$2 GPU_MAX_HW_QUEUES=$3:', round(d['value'], 2), 'frames/s', [round(x, 3) for x in d['repetitions']['ms_per_step_each']])
except Exception as e:
    print('groups=$1 launch=$2 GPU_MAX_HW_QUEUES=$3: failed', e, open('gpurun_out/r3_groups_bench.err').read()[-400:])
PY
done
