# round 2 profiles: kernel trace + stats of the bench command, memory-side traffic passes, SQ counters of the shipped GEMM variants and
# of the attention kernels.  Counter passes are their own runs (--kernel-trace + --pmc only).  Usage: bash tools/run_gpu_r2prof.sh <tag>
TAG=${1:-r02}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
REPO="$GRAFT_REPO_ROOT"; cd "$REPO"
CMD="python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-parity-path"
( cd /tmp; rm -rf /tmp/p_*; 
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -o kt -- $CMD > $REPO/gpurun_out/${TAG}_kt.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fs -o fs -- $CMD > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_wsz -o wsz -- $CMD > /dev/null 2>&1
  SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  SQ2="SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_WAVES"
  SQ3="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
  for p in 1 2 3; do
    eval C=\$SQ$p
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/p_g$p -o g -- python $REPO/tools/gemm_pmc.py > /dev/null 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/p_a$p -o a -- python $REPO/tools/attn_pmc.py 3 > /dev/null 2>&1
  done )
# bring the small CSVs back: stats + counter collections
mkdir -p gpurun_out/${TAG}_prof
for d in kt fs wsz g1 g2 g3 a1 a2 a3; do
  mkdir -p gpurun_out/${TAG}_prof/$d
  find /tmp/p_$d -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_prof/$d/ \;
  find /tmp/p_$d -name "*counter_collection.csv" -exec cp {} gpurun_out/${TAG}_prof/$d/ \;
done
du -sh gpurun_out/${TAG}_prof; ls gpurun_out/${TAG}_prof/*
