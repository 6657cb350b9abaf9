mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/r3f_tests.log; tail -5 gpurun_out/r3f_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
