export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -k "folded or norm2 or ffn_ln" -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs --no-ab --no-calibration"
for i in 1 2 3; do
  for L in libtoc3d_prev.so libtoc3d_gfx950.so; do
    TOC3D_LIB=$L $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value'],2), [round(x,4) for x in d['repetitions']['ms_per_step_each']])"
  done
done | tee gpurun_out/r4_batch_ab.txt
