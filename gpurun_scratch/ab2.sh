cd $GRAFT_REPO_ROOT
for n in base wgfull; do
  echo "== $n"; USC3D_LIB=$GRAFT_REPO_ROOT/build/ablate/$n.so timeout 200 python tools/conv_bench.py --sorted --reps 10 2>/dev/null | grep -E "128x96"
done
