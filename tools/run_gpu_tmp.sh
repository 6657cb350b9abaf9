cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for c in 0 1; do
echo "== normal stores, cold=$c"; tools/ubench/bin/gemm_timeline $c 1 2>&1 | grep -E "v16 bias" | cut -c1-330
echo "== contiguous-row stores (timing experiment), cold=$c"; tools/ubench/bin/gemm_timeline_contig $c 1 2>&1 | grep -E "v16 bias" | cut -c1-330
done | tee gpurun_out/r3_epi_contig.txt
