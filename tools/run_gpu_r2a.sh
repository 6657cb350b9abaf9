# round 2, first GPU pass: launch plans + bf16 parity + bench per launch mode
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_plan.py -q --tb=short -p no:cacheprovider -x 2>&1 | tail -25 > gpurun_out/r2a_plan.log; tail -8 gpurun_out/r2a_plan.log
timeout 1500 python -m pytest tests/test_gpu_parity_bf16.py -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r2a_parity.log; grep -E "^\[|passed|failed|Error|assert" gpurun_out/r2a_parity.log | cut -c1-400
timeout 600 python tools/ubench/host_issue_time.py gpurun_out/tune_r2.json > gpurun_out/r2a_host.log 2>&1; cat gpurun_out/r2a_host.log | tail -12
for mode in plan graph eager; do
  timeout 600 python bench.py --steps 100 --warmup 10 --launch $mode --tune-cache gpurun_out/tune_r2.json --no-cpu-baseline --no-breakdown > gpurun_out/r2a_bench_$mode.json 2> gpurun_out/r2a_bench_$mode.err
  python -c "import json;d=json.load(open('gpurun_out/r2a_bench_$mode.json'));print('$mode', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3),'ms')"
done
timeout 600 python bench.py --steps 100 --warmup 10 --launch plan --groups 3 --tune-cache gpurun_out/tune_r2.json --no-cpu-baseline --no-breakdown > gpurun_out/r2a_bench_plan_g3.json 2> gpurun_out/r2a_bench_plan_g3.err; python -c "import json;d=json.load(open('gpurun_out/r2a_bench_plan_g3.json'));print('plan g3', round(d['value'],1))"
timeout 600 python bench.py --steps 100 --warmup 10 --launch plan --groups 1 --tune-cache gpurun_out/tune_r2.json --no-cpu-baseline --no-breakdown > gpurun_out/r2a_bench_plan_g1.json 2> gpurun_out/r2a_bench_plan_g1.err; python -c "import json;d=json.load(open('gpurun_out/r2a_bench_plan_g1.json'));print('plan g1', round(d['value'],1))"
