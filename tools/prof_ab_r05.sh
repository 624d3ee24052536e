# kernel stats of the bench command under two switch settings: bash tools/prof_ab_r05.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-pab}
O=gpurun_out/$T; mkdir -p $O
prof() { name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/$name -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-reference-order --steps 6 --warmup 3 > $GRAFT_REPO_ROOT/$O/$name.log 2>&1)
  python tools/prof_summary.py $O/$name 140 > $O/$name.summary.txt
  rm -rf $O/$name
}
prof base USC3D_SORTED_CH=32 USC3D_WGRAD_BIG=0 USC3D_BN_TILE_ROWS=0
prof new  USC3D_SORTED_CH=32 USC3D_WGRAD_BIG=1 USC3D_BN_TILE_ROWS=4096
head -3 $O/base.summary.txt; head -3 $O/new.summary.txt
