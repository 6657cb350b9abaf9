mkdir -p gpurun_out
python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -k "reload" 2>&1 | tail -5
python bench.py --steps 30 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -30 gpurun_out/bench.err; cat gpurun_out/bench.json
python bench.py --steps 30 --warmup 5 --no-graph --no-cpu-baseline --no-breakdown > gpurun_out/bench_eager.json 2>> gpurun_out/bench.err; cat gpurun_out/bench_eager.json
REPO=$PWD; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof -o r01 -- python $REPO/bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-breakdown > $REPO/gpurun_out/prof.log 2>&1; cd $REPO; ls -R gpurun_out/prof | head; tail -3 gpurun_out/prof.log
