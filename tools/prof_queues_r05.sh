# kernel trace of the bench command, reported per hardware queue: bash tools/prof_queues_r05.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-pq}
O=gpurun_out/$T; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-zorder --steps 6 --warmup 3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
python tools/prof_summary.py $O/prof 70 > $O/summary.txt
python tools/stream_overlap.py $O/prof 45 > $O/overlap.txt
rm -rf $O/prof
cat $O/overlap.txt
