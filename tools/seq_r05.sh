# one step's kernel sequence at 150 k voxels (and optionally 20 k): bash tools/seq_r05.sh <tag> [voxels...]
cd $GRAFT_REPO_ROOT
T=${1:-seq}; shift
O=gpurun_out/$T; mkdir -p $O
for v in ${@:-150000}; do
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace$v -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-reference-order --steps 4 --warmup 3 --voxels $v > $GRAFT_REPO_ROOT/$O/trace$v.log 2>&1)
  python tools/step_kernel_sequence.py $O/trace$v > $O/seq$v.txt
  rm -rf $O/trace$v
  wc -l $O/seq$v.txt
done
