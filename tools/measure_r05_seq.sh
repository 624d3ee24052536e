# round-5: one step's kernel sequence (both queues) at 150 k and 20 k voxels + the default line.  bash tools/measure_r05_seq.sh <tag>
set -x
cd $GRAFT_REPO_ROOT
T=${1:-r05a}
O=gpurun_out/$T; mkdir -p $O
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -n 1 $O/bench.json | cut -c 1-400
for v in 150000 20000; do
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace$v -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-reference-order --steps 4 --warmup 3 --voxels $v > $GRAFT_REPO_ROOT/$O/trace$v.log 2>&1)
  python tools/step_kernel_sequence.py $O/trace$v > $O/seq$v.txt
  rm -rf $O/trace$v
  wc -l $O/seq$v.txt
done
