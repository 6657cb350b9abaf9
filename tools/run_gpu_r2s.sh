# round 2: 192x192 tile variants -- bit-identity test, sweep on the frame's shapes, bench with a fresh tile table
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -k "bit_identical or bit_stable or folded or conv3x3" 2>&1 | tail -3
VARIANTS=16,116,45,145,49,149,52,152,53 timeout 900 python tools/gemm_phased_check.py time 2>&1 | grep -E "^qkv|^proj|^w3" | cut -c1-330
for rep in 1 2; do
rm -f gpurun_out/tune_s.json
timeout 900 python bench.py --tune-cache gpurun_out/tune_s.json --no-cpu-baseline --no-parity-path > gpurun_out/r2s_bench_$rep.json 2> gpurun_out/r2s_bench_$rep.err
python -c "import json;d=json.load(open('gpurun_out/r2s_bench_$rep.json'));print(round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms', round(d['roofline']['frac'],4))"
done
grep -o "v52\|v152\|v53" gpurun_out/tune_s.json | sort | uniq -c; python -c "
import json; t=json.load(open('gpurun_out/tune_s.json'))['table']; print([ (k,v) for k,v in t if v in (52,53,152)])"
