mkdir -p gpurun_out
python -m pytest tests/test_gpu_attn_rot.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -25 > gpurun_out/r3d_rot.log; tail -5 gpurun_out/r3d_rot.log
python tools/ubench/attn_timeline.py > gpurun_out/r3_attn_timeline_v2.txt 2>&1; cut -c1-330 gpurun_out/r3_attn_timeline_v2.txt
python bench.py --no-cpu-baseline --no-batched --no-parity-path > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err; cut -c1-200 gpurun_out/r3d_bench.json; grep -E "qkv_rope|attention_rot|  sum" gpurun_out/r3d_bench.err | cut -c1-130
TOC3D_PREFETCH=0 python bench.py --no-cpu-baseline --no-batched --no-parity-path --no-breakdown 2>/dev/null | cut -c1-200
TOC3D_PREFETCH=96 python bench.py --no-cpu-baseline --no-batched --no-parity-path --no-breakdown 2>/dev/null | cut -c1-200
TOC3D_ATTN_ROT=0 python bench.py --no-cpu-baseline --no-batched --no-parity-path --no-breakdown 2>/dev/null | cut -c1-200
