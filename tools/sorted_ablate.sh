# Developer aid (GPU box): where does a coarse-level sorted-conv launch spend its time?  Builds ablation variants of the
# mask-sorted kernel (no reduction loop / no tile write-back / neither) and times the 507-, 2 222- and 9 402-row layers.
#   bash tools/sorted_ablate.sh > gpurun_out/sorted_ablate.txt
cd $GRAFT_REPO_ROOT
bash tools/build_ablate.sh s_loop "-DUSC_ABLATE_SORTED_LOOP" spconv_sorted.hip >/dev/null
bash tools/build_ablate.sh s_store "-DUSC_ABLATE_SORTED_STORE" spconv_sorted.hip >/dev/null
bash tools/build_ablate.sh s_both "-DUSC_ABLATE_SORTED_LOOP -DUSC_ABLATE_SORTED_STORE" spconv_sorted.hip >/dev/null
for shape in 16:256x256 8:128x128 8:256x256 4:128x128 4:64x64; do
  for v in full s_loop s_store s_both; do
    lib=""; [ $v != full ] && lib="USC3D_LIB=build/ablate/$v.so"
    echo -n "$shape $v: "
    env $lib python tools/conv_bench.py --sorted --only $shape --reps 30 2>/dev/null | grep -E "^ +[0-9]+ +[0-9]+ " | head -1
    for G in 1 27; do
      echo -n "$shape $v G=$G NB=4: "
      env $lib USC3D_SORTED_TUNE=1 USC3D_SORTED_NB=4 USC3D_SORTED_G=$G python tools/conv_bench.py --sorted --only $shape --reps 30 2>/dev/null | grep -E "^ +[0-9]+ +[0-9]+ " | head -1
    done
  done
done
