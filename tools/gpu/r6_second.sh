# round 6, second GPU pass: x3 phased tiles (parity), fp32x3 bench with the capped rotating epilogue, then an in-place pass of the fp32x3 table over the phased variants
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tuned
timeout 900 python -m pytest tests/test_gpu_attn_rot.py tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "planes or x3" 2>&1 | tail -15 | tee gpurun_out/r6_planes_tests.txt
timeout 600 python bench.py --precision fp32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-batched --no-parity-path --no-other-configs --no-ab 2>gpurun_out/r6_bench_x3.err | tail -1 > gpurun_out/r6_bench_x3.json
python -c "
import json; d=json.loads(open('gpurun_out/r6_bench_x3.json').read()); print('fp32x3', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"
grep -n "qkv_rope \|attention_rot \|toc3d_linear_fused " gpurun_out/r6_bench_x3.err | head
T=toc3d_faster_320x800_fp32x3.json
TOC3D_TUNE_CANDS=60,61,62,63,160,161,162,163 timeout 2400 python tools/tune_in_place.py toc3d_amd/tuned/$T gpurun_out/tuned/$T toc3d_faster 30 tiles 320x800 fp32x3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_retune_x3_phased.txt | tail -40
