set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
# 1. default bench line (with cpu baseline)
python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err
# 2. step time vs scene size
for v in 20000 80000 150000 300000; do python bench.py --no-cpu-baseline --voxels $v --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['config']['voxels_per_scene'], round(r['ms_per_step'],2))"; done > gpurun_out/r02/step_vs_size.txt
for v in 20000 150000; do python bench.py --no-cpu-baseline --no-prefetch --voxels $v --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('no-prefetch', r['config']['voxels_per_scene'], round(r['ms_per_step'],2))"; done >> gpurun_out/r02/step_vs_size.txt
# 3. kernel stats of the bench command
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02/prof.log 2>&1)
python tools/prof_summary.py gpurun_out/r02/prof 80 > gpurun_out/r02/kernel_stats_summary.txt
cp $(ls gpurun_out/r02/prof/*/*kernel_stats.csv | head -1) gpurun_out/r02/kernel_stats.csv
# 4. PMC traffic passes (separate runs, counters only with kernel-trace)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/pmc_summary.py gpurun_out/r02/pmc_fetch gather_gemm > gpurun_out/r02/pmc_traffic_raw.txt
python tools/pmc_summary.py gpurun_out/r02/pmc_fetch wgrad >> gpurun_out/r02/pmc_traffic_raw.txt
python tools/pmc_summary.py gpurun_out/r02/pmc_write gather_gemm >> gpurun_out/r02/pmc_traffic_raw.txt
python tools/pmc_summary.py gpurun_out/r02/pmc_write wgrad >> gpurun_out/r02/pmc_traffic_raw.txt
rm -rf gpurun_out/r02/pmc_fetch gpurun_out/r02/pmc_write gpurun_out/r02/prof
# 5. HBM report, ncut bench, conv per shape
python tools/hbm_report.py > gpurun_out/r02/hbm_bound_kernels.txt 2>/dev/null
python bench.py --mode ncut --scenes 2 > gpurun_out/r02/bench_ncut.json 2>/dev/null
python bench.py --mode ncut --scenes 1 --no-cpu-baseline > gpurun_out/r02/bench_ncut_k1.json 2>/dev/null
USC3D_PROF_SHAPES=1 python tools/conv_report.py > gpurun_out/r02/conv_per_shape.txt 2>/dev/null
ls -la gpurun_out/r02
