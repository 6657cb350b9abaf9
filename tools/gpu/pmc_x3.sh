# SQ counters of the fp32x3 frame's kernels (two passes; tools/summarize_pmc.py): the planes attention's LDS conflicts, MFMA busy, waits
TAG=${TAG:-r06_fp32x3}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out"
BENCH="python bench.py --precision fp32x3 --steps 4 --warmup 2 --reps 1 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs --no-ab --no-calibration"
rm -rf $OUT/sq1 $OUT/sq2
timeout 900 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/sq1 -o sq1 --output-format csv -- $BENCH > /dev/null 2> $OUT/sq1.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVES -d $OUT/sq2 -o sq2 --output-format csv -- $BENCH > /dev/null 2> $OUT/sq2.err
python tools/summarize_pmc.py $OUT/${TAG}_pmc_summary.txt "rocprofv3 --kernel-trace --pmc <counters> -- $BENCH (two SQ passes; per-launch averages per kernel instantiation; raw counters)" $OUT/sq1 $OUT/sq2 2>&1 | tail -2
rm -rf $OUT/sq1 $OUT/sq2
grep -n "attn_rot_x3" $OUT/${TAG}_pmc_summary.txt | cut -c1-400 | head -12
