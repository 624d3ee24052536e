cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "conv or sorted or large_map or families or adjoint or 150k" > gpurun_out/r06_t5.log 2>&1; tail -3 gpurun_out/r06_t5.log
bash tools/build_ablate.sh phase "-DUSC_PHASE_STATS" >/dev/null 2>&1
USC3D_LIB=build/ablate/phase.so python tools/compact_phase.py
python tools/conv_bench.py --only 1:96x96
python tools/conv_bench.py --only 1:128x96
python tools/conv_bench.py --only 2:96x96
bash tools/ab.sh -r 3 new: 
for t in 4 8 16; do s=$(date +%s); OMP_NUM_THREADS=$t python -m pytest "tests/test_gpu_step_parity.py::test_config3_three_step_loss_trajectory" -q -p no:cacheprovider 2>&1 | tail -1; echo "threads $t: $(( $(date +%s) - s )) s"; done
