for v in "" cw16_tm196 cw16_tm192 cw8_tm196; do
  if [ -z "$v" ]; then L=""; else L="USC3D_LIB=build/ablate/$v.so"; fi
  for o in 1:96x96 2:96x96 1:128x96; do
    echo -n "variant=${v:-base} "; env $L python tools/conv_bench.py --sorted --reps 20 --only $o 2>/dev/null | grep -E "^\s+[0-9]+ +[0-9]+ +[0-9]+x" | head -1
  done
done
