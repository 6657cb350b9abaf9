# round 4: stats-out combine without the extra barrier on single-buffer tiles: chain test detail (EXPERIMENTAL=1 library), fold tests, per-launch cost, library A/B
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
TOC3D_LIB=libtoc3d_gfx950_exp.so timeout 900 python -m pytest tests/test_gpu_chain.py -k "equals_separate" -q --tb=short -p no:cacheprovider 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r4_chain_detail.txt; tail -25 gpurun_out/r4_chain_detail.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_plan.py -k "folded or norm2 or ffn_ln or bit_stable" -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
timeout 600 python tools/ubench/fold_cost.py > gpurun_out/r4_fold_cost5.txt 2>&1; tail -30 gpurun_out/r4_fold_cost5.txt
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs --no-ab --no-calibration"
for i in 1 2 3 4 5; do
  for L in libtoc3d_prev.so libtoc3d_gfx950.so; do
    TOC3D_LIB=$L $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value'],2), [round(x,4) for x in d['repetitions']['ms_per_step_each']])"
  done
done | tee gpurun_out/r4_barrier_ab.txt
