# kernel trace of the shipped default (norm2 fold on): profiles/r03_kernel_stats_norm2_fold.csv and r03_where_time_goes_norm2_fold.txt come from it
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out"
rm -rf $OUT/kt2
timeout 280 rocprofv3 --kernel-trace --stats -d $OUT/kt2 -o kt --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs > $OUT/trace2_bench.json 2> $OUT/trace2.err
for f in $(find $OUT/kt2 -mindepth 2 -name "kt_*.csv"); do cp $f $OUT/kt2/; done
find $OUT/kt2 -mindepth 1 -type d -exec rm -rf {} + 2>/dev/null
python tools/frame_timeline.py $OUT/kt2/kt_kernel_trace.csv -3 -5 > $OUT/trace2_frames.txt 2>&1
head -12 $OUT/trace2_frames.txt | cut -c1-200; cat $OUT/trace2_bench.json | cut -c1-300
