mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
REPO=$PWD
python bench.py --steps 5 --warmup 3 --groups 1 --no-cpu-baseline --no-breakdown --tune-cache /tmp/tune.json > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 10 --warmup 3 --groups 1 --no-cpu-baseline --no-breakdown --tune-cache /tmp/tune.json"
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/kt -o kt -- $CMD > $REPO/gpurun_out/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/fs -o fs -- $CMD > $REPO/gpurun_out/fs.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/wsz -o wsz -- $CMD > $REPO/gpurun_out/wsz.log 2>&1
cd $REPO; cp /tmp/tune.json gpurun_out/tune.json; tail -1 gpurun_out/kt.log | cut -c1-200
