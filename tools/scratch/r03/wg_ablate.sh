for v in "" wg_epi wg_loop wg_both; do
  if [ -z "$v" ]; then L=""; else L="USC3D_LIB=$PWD/build/ablate/$v.so"; fi
  echo "== ${v:-baseline}"
  env $L USC3D_PROF_SHAPES=1 python tools/conv_report.py 2>/dev/null | grep -E "wgrad_full_kernel<2, 4> \[n=(13689|59994) cin=(256|128) cout=(256|128) K=27|wgrad_full_kernel<3, 3> \[n=4011228|total conv"
done
