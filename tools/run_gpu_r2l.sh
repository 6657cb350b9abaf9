# round 2: norm2-fold op test (fixed), fold / prefetch e2e test, A/B of the prefetch riding on the attention launches
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -q --tb=short -p no:cacheprovider -x -k "folded or attention" -s 2>&1 | grep -E "^\[|passed|failed|Error|assert" | cut -c1-200 | tail -14
cp toc3d_amd/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tune_pf.json
run() { TOC3D_PREFETCH=$2 timeout 600 python bench.py --steps 100 --warmup 10 --tune-cache gpurun_out/tune_pf.json --no-cpu-baseline --no-breakdown --no-parity-path > gpurun_out/r2l_$1.json 2> gpurun_out/r2l_$1.err
  python -c "import json;d=json.load(open('gpurun_out/r2l_$1.json'));print('$1', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms')"; }
run warm 0
for rep in 1 2 3; do run pf0_$rep 0; run pf32_$rep 32; run pf96_$rep 96; run pf256_$rep 256; done
