# round 6: library A/B of the fast SiLU on the fp32x3 leg, then the committed profiles of the round: bf16 headline (TAG=r06) and the fp32x3 kernel stats / frame cut
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT="$GRAFT_REPO_ROOT/gpurun_out"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -x -k "x3 or planes or swiglu or vitl_fp32" 2>&1 | grep -v amdgpu.ids | tail -5
N=3 BENCH_ARGS="--precision fp32x3" bash tools/gpu/lib_ab.sh 2>&1 | tail -9
TAG=r06 bash tools/gpu/profile.sh 2>&1 | tail -45
X3="python bench.py --precision fp32x3 --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs --no-ab --no-calibration"
rm -rf $OUT/kt
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $X3 > $OUT/prof_bench_x3.json 2> $OUT/prof_x3.err
for f in $(find $OUT/kt -mindepth 2 -name "kt_*.csv"); do cp $f $OUT/kt/; done
python tools/frame_timeline.py $OUT/kt/kt_kernel_trace.csv > $OUT/r06_fp32x3_where_time_goes.txt 2>&1; head -30 $OUT/r06_fp32x3_where_time_goes.txt
python - <<'PY'
import subprocess, sys, os, shutil
# kernel stats of the fp32x3 run through the same summariser (its PMC part needs passes we did not take for this leg: only the stats file is kept)
import re
src = open('tools/summarize_prof.py').read()
cut = src.index('def pmc(dirname, prefix, counter):')
code = src[:cut].replace('tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"', 'tag = "r06_fp32x3"').replace('cmd = sys.argv[3] if len(sys.argv) > 3 else', 'cmd = "python bench.py --precision fp32x3 --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs --no-ab --no-calibration" if True else')
exec(compile(code, 'summ', 'exec'))
shutil.copy('profiles/r06_fp32x3_kernel_stats.csv', 'gpurun_out/r06_fp32x3_kernel_stats.csv')
PY
find $OUT/kt -mindepth 1 -type d -exec rm -rf {} + 2>/dev/null
find $OUT/kt -name "*kernel_trace.csv" -size +50M -delete
tail -1 $OUT/prof_bench_x3.json | cut -c1-200
