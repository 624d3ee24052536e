cd $GRAFT_REPO_ROOT
time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4
time python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 400 gpurun_out/bench_default.json
