cd $GRAFT_REPO_ROOT
O=gpurun_out/r03pmc; mkdir -p $O
for shp in 16:256x256 8:128x128 4:64x64; do
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/p_$shp -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --sorted --only $shp --reps 5 > $GRAFT_REPO_ROOT/$O/log_$shp.txt 2>&1)
  echo "== $shp"; grep -E "^ +[0-9]+ " $O/log_$shp.txt | tail -2
  (python tools/pmc_summary.py $O/p_$shp sorted; python tools/pmc_summary.py $O/p_$shp wgrad_full; python tools/pmc_summary.py $O/p_$shp reduce) > $O/pmc_$shp.txt
  rm -rf $O/p_$shp
done
