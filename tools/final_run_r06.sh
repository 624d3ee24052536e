cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06f
s=$(date +%s)
python -m pytest tests -m gpu -q --durations=30 -p no:cacheprovider > gpurun_out/r06f/tests.log 2>&1; tail -2 gpurun_out/r06f/tests.log
echo "gpu suite: $(( $(date +%s) - s )) s" | tee gpurun_out/r06f/suite_seconds.txt
bash tools/measure_r06.sh r06f > gpurun_out/r06f/measure.log 2>&1
tail -3 gpurun_out/r06f/measure.log
