cd $GRAFT_REPO_ROOT
for n in base noflush noticket normw noA noB noAB noABflush kda2 kda4 kdb2 krb4; do
  echo -n "$n: "; USC3D_LIB=$GRAFT_REPO_ROOT/build/ablate/$n.so timeout 200 python tools/conv_bench.py --sorted --only 1:96x96 --reps 10 2>/dev/null | grep "96x96" | head -1
done
