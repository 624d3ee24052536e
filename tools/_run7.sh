cd $GRAFT_REPO_ROOT
for v in "" build/ablate/afirst.so build/ablate/bfirst_da2.so; do echo "== lib ${v:-main (B first)}"; USC3D_LIB=$v python tools/conv_bench.py --only 1:96x96 | grep 148564; USC3D_LIB=$v python tools/conv_bench.py --only 2:96x96 | grep "40421   96"; done
bash tools/ab.sh -r 3 bfirst: afirst:USC3D_LIB=build/ablate/afirst.so
bash tools/prof_timeline.sh r06_tl_forcedist X=1 --force-dist
cat gpurun_out/r06_tl_forcedist/timeline_sections.txt
