# rocprofv3 kernel stats of the bench command with the weight-gradient lane OFF: the dominant kernel's own duration (no lane kernel beside it)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && USC3D_WGRAD_LANE_MAX_ROWS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-zorder > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
python tools/prof_summary.py $O/prof 40 > $O/kernel_stats_lane_off_summary.txt
rm -rf $O/prof
head -8 $O/kernel_stats_lane_off_summary.txt
python tools/op_census.py --top 60 > $O/op_census.txt 2>&1; tail -70 $O/op_census.txt
