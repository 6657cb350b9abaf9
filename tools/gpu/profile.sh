# profiles: kernel trace + stats of the bench command, the frame timeline cut, and the two PMC passes for the GEMM's memory-side traffic
# usage: TAG=r05 bash tools/gpu/profile.sh   (TAG names the files written for profiles/: <TAG>_kernel_stats.csv, <TAG>_gemm_hbm_traffic.json, <TAG>_pmc_summary.txt, <TAG>_where_time_goes.txt)
TAG=${TAG:-r05}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out"
BENCH="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs --no-ab --no-calibration"
rm -rf $OUT/kt $OUT/fs $OUT/wsz
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $BENCH > $OUT/prof_bench.json 2> $OUT/prof.err
for f in $(find $OUT/kt -mindepth 2 -name "kt_*.csv"); do cp $f $OUT/kt/; done
python tools/frame_timeline.py $OUT/kt/kt_kernel_trace.csv > $OUT/${TAG}_where_time_goes.txt 2>&1; head -30 $OUT/${TAG}_where_time_goes.txt
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fs -o fs --output-format csv -- $BENCH --steps 4 --warmup 2 --reps 1 > /dev/null 2> $OUT/fs.err
for f in $(find $OUT/fs -mindepth 2 -name "fs_*.csv"); do cp $f $OUT/fs/; done
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/wsz -o wsz --output-format csv -- $BENCH --steps 4 --warmup 2 --reps 1 > /dev/null 2> $OUT/wsz.err
for f in $(find $OUT/wsz -mindepth 2 -name "wsz_*.csv"); do cp $f $OUT/wsz/; done
rm -rf $OUT/sq1 $OUT/sq2
timeout 900 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/sq1 -o sq1 --output-format csv -- $BENCH --steps 4 --warmup 2 --reps 1 > /dev/null 2> $OUT/sq1.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVES -d $OUT/sq2 -o sq2 --output-format csv -- $BENCH --steps 4 --warmup 2 --reps 1 > /dev/null 2> $OUT/sq2.err
python tools/summarize_pmc.py $OUT/${TAG}_pmc_summary.txt "rocprofv3 --kernel-trace --pmc <counters> -- $BENCH --steps 4 --warmup 2 --reps 1 (two SQ passes; per-launch averages per kernel instantiation; raw counters)" $OUT/sq1 $OUT/sq2 2>&1 | tail -2
find $OUT/kt $OUT/fs $OUT/wsz $OUT/sq1 $OUT/sq2 -mindepth 1 -type d -exec rm -rf {} + 2>/dev/null
rm -rf $OUT/sq1 $OUT/sq2
find $OUT/kt $OUT/fs $OUT/wsz -name "*kernel_trace.csv" -size +50M -delete
python tools/summarize_prof.py $TAG gpurun_out "$BENCH" 2>&1 | tail -3
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_gemm_hbm_traffic.json $OUT/ 2>/dev/null
tail -2 $OUT/prof_bench.json | cut -c1-300
