cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h
mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 4 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
python tools/stock_kernel_context.py $O/prof 120 > $O/stock_context.txt; python tools/step_kernel_sequence.py $O/prof > $O/step_sequence.txt
rm -rf $O/prof
head -5 $O/stock_context.txt
