# SQ counters of the tile-compacted conv kernel (96->96 on the 148 564-row map, forward + input gradient) for the shipped
# build and two developer builds: no operand loads, and no operand loads + no flush ("skeleton": prologue, compaction,
# matrix-core issue, tile write-back).  Usage (GPU box): bash tools/compact_sq.sh > out.txt
cd $GRAFT_REPO_ROOT
for v in "" c_noloads c_skeleton; do
  if [ -z "$v" ]; then L=""; else L="USC3D_LIB=$GRAFT_REPO_ROOT/build/ablate/$v.so"; fi
  D=gpurun_out/csq_${v:-base}; rm -rf $D
  (cd /tmp && export TMPDIR=/tmp && env $L rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$D -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --sorted --reps 5 --only 1:96x96 > /dev/null 2>&1)
  echo "## variant ${v:-shipped}"
  python tools/pmc_summary.py $D gather_gemm_compact
  rm -rf $D
done
