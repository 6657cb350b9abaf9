# round 2: implicit-GEMM neck + flash 128-query switch + defaults (prefetch 192): tests, attention timing, bench A/B
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -q --tb=short -p no:cacheprovider -x -k "conv3x3 or neck or im2col or loop" -s 2>&1 | grep -E "^\[|passed|failed|Error|assert" | cut -c1-200 | tail -12
echo "--- flash QM=1"; python tools/attn_pmc.py 20 2>&1 | grep flash
echo "--- flash QM=2"; TOC3D_ATTN_QM=2 python tools/attn_pmc.py 20 2>&1 | grep flash
cp toc3d_amd/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tune_n.json
run() { TOC3D_ATTN_QM=$2 timeout 600 python bench.py --steps 100 --warmup 10 --tune-cache gpurun_out/tune_n.json --no-cpu-baseline --no-breakdown --no-parity-path > gpurun_out/r2n_$1.json 2> gpurun_out/r2n_$1.err
  python -c "import json;d=json.load(open('gpurun_out/r2n_$1.json'));print('$1', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms')"; }
run warm 1
for rep in 1 2 3; do run qm1_$rep 1; run qm2_$rep 2; done
timeout 600 python bench.py --steps 50 --warmup 10 --tune-cache gpurun_out/tune_n.json --no-cpu-baseline --no-parity-path > gpurun_out/r2n_breakdown.json 2> gpurun_out/r2n_breakdown.err; head -22 gpurun_out/r2n_breakdown.err
