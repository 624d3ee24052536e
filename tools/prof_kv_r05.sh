# kernel traces of the bench command with the decoder key-preparation stream on / off: bash tools/prof_kv_r05.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-pkv}
O=gpurun_out/$T; mkdir -p $O
prof() { name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/$name -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-zorder --steps 6 --warmup 3 > $GRAFT_REPO_ROOT/$O/$name.log 2>&1)
  python tools/prof_summary.py $O/$name 60 > $O/$name.summary.txt
  python tools/stream_overlap.py $O/$name 14 > $O/$name.overlap.txt
  python tools/stream_timeline.py $O/$name > $O/$name.timeline.txt
  rm -rf $O/$name
}
prof on  USC3D_KV_SIDE_STREAM=1
head -4 $O/on.overlap.txt
