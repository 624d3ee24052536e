cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in new old; do
if [ $v = old ]; then export USC3D_LIB=$GRAFT_REPO_ROOT/gpurun_scratch/lib_old.so; else unset USC3D_LIB; fi
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$v', round(r['ms_per_step'],2), round(r['roofline']['all_conv_kernels']['usc::gather_gemm_sorted_kernel']['ms'],3))"
done; done
unset USC3D_LIB
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mask_sorted or conv3 or strided or edge_sizes" 2>&1 | tail -2
