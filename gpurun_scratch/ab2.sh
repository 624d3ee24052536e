cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02c/$tag -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --sorted --only 1:96x96 --reps 5 > $GRAFT_REPO_ROOT/gpurun_out/r02c/$tag.log 2>&1)
  for k in gather_gemm_compact wgrad_full; do python tools/pmc_summary.py gpurun_out/r02c/$tag $k; done >> gpurun_out/r02c/pmc_conv96x96_s1.txt
  rm -rf gpurun_out/r02c/$tag
done
cat gpurun_out/r02c/pmc_conv96x96_s1.txt | cut -c1-120
tail -3 gpurun_out/r02c/SQ_WAVE_CYCLES.log | cut -c1-200
