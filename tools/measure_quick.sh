# quick mid-round measurement: bench line, kernel stats of the bench command, conv per shape
# usage (on the GPU box): bash tools/measure_quick.sh <tag>
set -x
cd $GRAFT_REPO_ROOT
T=${1:-r03a}
O=gpurun_out/$T
mkdir -p $O
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
python tools/prof_summary.py $O/prof 90 > $O/kernel_stats_summary.txt
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/prof
USC3D_PROF_SHAPES=1 python tools/conv_report.py > $O/conv_per_shape.txt 2>/dev/null
cat $O/bench.json
