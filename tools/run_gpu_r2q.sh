# round 2: gathered residual (no f32 shortcut copy): op tests, e2e parity, same-box A/B
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -x -k "gather or folded or bit_identical or residual" 2>&1 | tail -4
cp toc3d_amd/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tune_q.json
run() { TOC3D_GATHERED_RES=$2 timeout 600 python bench.py --steps 100 --warmup 10 --tune-cache gpurun_out/tune_q.json --no-cpu-baseline --no-breakdown --no-parity-path > gpurun_out/r2q_$1.json 2> gpurun_out/r2q_$1.err
  python -c "import json;d=json.load(open('gpurun_out/r2q_$1.json'));print('$1', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms')"; }
run warm 1
for rep in 1 2 3; do run copy_$rep 0; run gathered_$rep 1; done
timeout 2400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_bf16.py tests/test_gpu_plan.py -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
