# round 2: prefetch riding on the attention launches, prefetch workgroups dispatched first: A/B over the workgroup count
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -q --tb=short -p no:cacheprovider -x -k "folded or attention or swiglu" 2>&1 | tail -3
cp toc3d_amd/tuned/toc3d_faster_320x800_bf16.json gpurun_out/tune_pf.json
run() { TOC3D_PREFETCH=$2 timeout 600 python bench.py --steps 100 --warmup 10 --tune-cache gpurun_out/tune_pf.json --no-cpu-baseline --no-breakdown --no-parity-path > gpurun_out/r2m_$1.json 2> gpurun_out/r2m_$1.err
  python -c "import json;d=json.load(open('gpurun_out/r2m_$1.json'));print('$1', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms')"; }
run warm 0
for rep in 1 2 3; do run pf0_$rep 0; run pf128_$rep 128; run pf256_$rep 256; run pf512_$rep 512; run pf1024_$rep 1024; done
