cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
s=$(date +%s)
python -m pytest tests -m gpu -q --durations=40 -p no:cacheprovider > gpurun_out/r06_tests.log 2>&1; tail -3 gpurun_out/r06_tests.log
echo "gpu suite: $(( $(date +%s) - s )) s"
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; tail -c 3000 gpurun_out/r06_bench.json
bash tools/prof_timeline.sh r06_tl
cat gpurun_out/r06_tl/timeline_sections.txt | head -40
