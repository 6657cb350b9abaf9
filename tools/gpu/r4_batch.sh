mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_e2e.py -k "two_frames_batch" -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
TOC3D_KEEP_TABLE=1 timeout 2400 python tools/make_tuned_tables.py gpurun_out/tuned "toc3d_faster:320x800:bf16:1,2,4,8" 2>&1 | grep -v amdgpu.ids | tail -3
cp gpurun_out/tuned/toc3d_faster_320x800_bf16.json toc3d_amd/tuned/toc3d_faster_320x800_bf16.json
for extra in "--sequential-frames" ""; do
  timeout 900 python bench.py --frames-total 8 $extra --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs --no-ab --no-calibration 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames-total 8 $extra:', round(d['value'],1), 'frames/s', d['repetitions']['ms_per_step_each'], d['config']['frames_per_forward'])"
done | tee gpurun_out/r4_batch.txt
for B in 2 4; do
  timeout 900 python bench.py --frames-total $B --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs --no-ab --no-calibration 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames-total $B:', round(d['value'],1), 'frames/s', d['repetitions']['ms_per_step_each'], d['config']['frames_per_forward'])"
done | tee -a gpurun_out/r4_batch.txt
