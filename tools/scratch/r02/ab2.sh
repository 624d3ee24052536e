cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02d/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02d/prof.log 2>&1)
python tools/idle_gaps.py gpurun_out/r02d/prof > gpurun_out/r02d/idle_gaps.txt 2>&1
cat gpurun_out/r02d/idle_gaps.txt | cut -c1-220
rm -rf gpurun_out/r02d/prof
