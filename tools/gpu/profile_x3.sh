# kernel trace + stats of the fp32x3 bench command and the frame cut (TAG names the files: <TAG>_kernel_stats.csv via tools/summarize_prof.py on the merged kt/ directory, <TAG>_where_time_goes.txt)
TAG=${TAG:-r06_fp32x3}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out"
X3="python bench.py --precision fp32x3 --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs --no-ab --no-calibration"
rm -rf $OUT/kt
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $X3 > $OUT/prof_bench_x3.json 2> $OUT/prof_x3.err
for f in $(find $OUT/kt -mindepth 2 -name "kt_*.csv"); do cp $f $OUT/kt/; done
python tools/frame_timeline.py $OUT/kt/kt_kernel_trace.csv > $OUT/${TAG}_where_time_goes.txt 2>&1; head -24 $OUT/${TAG}_where_time_goes.txt
find $OUT/kt -mindepth 1 -type d -exec rm -rf {} + 2>/dev/null
tail -1 $OUT/prof_bench_x3.json | cut -c1-160
