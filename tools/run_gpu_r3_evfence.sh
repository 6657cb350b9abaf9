# A/B of the event flags behind the launch plan's cross-lane edges (TOC3D_EVENT_FENCE = 0 default / 1 release-to-device / 2 no system fence), then the
# launch-plan tests under the most aggressive setting
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs"
for round in 1 2; do
  for m in 0 2 1; do
    TOC3D_EVENT_FENCE=$m timeout 60 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fence $m round $round:', round(d['value'],1), 'frames/s', [round(x,3) for x in d['repetitions']['ms_per_step_each']])"
  done
done | tee gpurun_out/evfence.txt
TOC3D_EVENT_FENCE=2 timeout 100 python -m pytest tests/test_gpu_plan.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4 | tee -a gpurun_out/evfence.txt
