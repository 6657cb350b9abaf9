# round-3 counters: memory-side traffic of the GEMM launches (FETCH_SIZE / WRITE_SIZE, separate passes), SQ counters of the GEMM and attention kernels,
# kernel stats of the strict-parity fp32 path and of the bf16 x 3 path.  Every pass = rocprofv3 --kernel-trace --pmc <counters> only.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out"
CMD="python bench.py --steps 4 --warmup 2 --reps 1 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs"
for pass in "fs FETCH_SIZE" "wsz WRITE_SIZE" "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"; do
  set -- $pass; d=$1; shift
  rm -rf $OUT/$d
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$d -o $d --output-format csv -- $CMD > $OUT/r3_pmc_$d.json 2> $OUT/r3_pmc_$d.err
  for f in $(find $OUT/$d -mindepth 2 -name "${d}_*.csv"); do cp $f $OUT/$d/; done
  find $OUT/$d -mindepth 1 -type d -exec rm -rf {} + 2>/dev/null
  find $OUT/$d -name "*kernel_trace.csv" -size +30M -delete
  ls $OUT/$d | head -5
done
for prec in fp32 fp32x3; do
  rm -rf $OUT/kt_$prec
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_$prec -o kt --output-format csv -- python bench.py --precision $prec --steps 3 --warmup 2 --reps 1 --no-cpu-baseline --no-breakdown --no-batched --no-parity-path --no-other-configs > $OUT/r3_kt_$prec.json 2> $OUT/r3_kt_$prec.err
  for f in $(find $OUT/kt_$prec -mindepth 2 -name "kt_*.csv"); do cp $f $OUT/kt_$prec/; done
  find $OUT/kt_$prec -mindepth 1 -type d -exec rm -rf {} + 2>/dev/null
  find $OUT/kt_$prec -name "*kernel_trace.csv" -delete
  tail -c 300 $OUT/r3_kt_$prec.json; echo
done
