# memory-side traffic of the wide GEMMs by XCD order (tools/gemm_traffic.py): four rocprofv3 --pmc passes (TCC has 4 slots per pass), then the summary
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/traffic"
rm -rf $OUT; mkdir -p $OUT
i=0
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o p$i --output-format csv -- python tools/gemm_traffic.py run > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log
done
python tools/gemm_traffic.py summarize $OUT/../r05/gemm_traffic_breakdown.txt $OUT/p1 $OUT/p2 $OUT/p3
find $OUT -name "*.csv" -size +20M -delete
